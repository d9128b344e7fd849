"""Emulated k_boxh + k_boxv (two-pass box) vs oracle: 64bpp class at ratios >= 12 and every 128bpp-class box job."""
import sys, random
sys.path.insert(0, "tests")
import numpy as np
from helpers import *
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 3); bad=0; n=0; seen={}
N=int(sys.argv[2]) if len(sys.argv) > 2 else 100
for it in range(N):
    w_out=random.choice([1,2,3,7,40,64,65,100]); h_out=random.choice([1,2,3,5,9])
    rx=random.uniform(8.5,40); ry=random.uniform(8.5,40)
    w_in=min(int(w_out*rx)+1, 4000); h_in=min(int(h_out*ry)+1, 400)
    t_in=random.randrange(10); t_out=random.randrange(10); srgb=random.choice([0,0,1])
    sin=w_in*bpp(t_in)+random.choice([0,0,1,3,4,16]); sout=w_out*bpp(t_out)+random.choice([0,0,1,4])
    job=plan_job(t_in,w_in,h_in,t_out,w_out,h_out,srgb)
    if job.filter_h!=3 or job.filter_v!=3: continue
    src=make_input(t_in,w_in,h_in,seed=it,valid=random.choice([True,False]),stride=sin,pad=0)
    first=random.randint(0,h_out-1); nr=random.randint(1,h_out-first)
    a=oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout,first,nr)
    b,k=emu_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout,first,nr)
    key=(k,job.wide)
    seen[key]=seen.get(key,0)+1
    n+=1
    if not np.array_equal(a,b):
        bad+=1; print("MISMATCH types",t_in,t_out,"srgb",srgb,(w_in,h_in),(w_out,h_out),sin,sout,first,nr,"class",k,"wide",job.wide)
print("cases",n,"bad",bad,"(class,wide):",seen)
