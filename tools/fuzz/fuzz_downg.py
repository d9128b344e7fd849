"""Emulated k_lind vs oracle: 128bpp-class minification, any types, srgb on/off, any alignment."""
import sys, random
sys.path.insert(0, "tests")
import numpy as np
from helpers import *
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 3); bad=0; n=0; seen={}
N=int(sys.argv[2]) if len(sys.argv) > 2 else 150
for it in range(N):
    w_out=random.choice([1,2,3,7,40,64,65,130,256,300]); h_out=random.choice([1,2,5,17,33])
    rx=random.uniform(1.01,7.99); ry=random.uniform(1.01,7.99)
    w_in=max(2,min(int(w_out*rx)+random.randint(0,1), 8*w_out-1)); h_in=max(2,min(int(h_out*ry)+random.randint(0,1), 8*h_out-1))
    if w_in<=w_out or h_in<=h_out: continue
    t_in=random.randrange(10); t_out=random.randrange(10); srgb=random.choice([0,1,1])
    sin=w_in*bpp(t_in)+random.choice([0,0,4,16,1,3]); sout=w_out*bpp(t_out)+random.choice([0,0,1,4,16])
    src=make_input(t_in,w_in,h_in,seed=it,valid=random.choice([True,False]),stride=sin,pad=0)
    job=plan_job(t_in,w_in,h_in,t_out,w_out,h_out,srgb)
    if not job.wide: continue
    first=random.randint(0,h_out-1); nr=random.randint(1,h_out-first)
    a=oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout,first,nr)
    b,k=emu_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout,first,nr)
    key=(k,job.premul16,job.linear)
    seen[key]=seen.get(key,0)+1
    n+=1
    if not np.array_equal(a,b):
        bad+=1; print("MISMATCH types",t_in,t_out,"srgb",srgb,(w_in,h_in),(w_out,h_out),sin,sout,first,nr,"class",k,"p16",job.premul16,"lin",job.linear,"halv",job.halvings_h,job.halvings_v)
print("cases",n,"bad",bad,"(class,p16,linear):",seen)
