"""Emulated k_tile vertical-copy specialisation vs oracle: width-only minification (h_in == h_out)."""
import sys, random
sys.path.insert(0, "tests")
import numpy as np
from helpers import *
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 3); bad=0; n=0; seen={}
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 120):
    w_out=random.choice([1,7,64,65,130,200]); h=random.choice([1,2,5,17,40])
    w_in=max(2,min(int(w_out*random.uniform(1.01,7.99))+1, 8*w_out-1))
    if w_in<=w_out: continue
    t_in=random.randrange(10); t_out=random.randrange(10)
    sin=w_in*bpp(t_in)+random.choice([0,0,1,3,4,16]); sout=w_out*bpp(t_out)+random.choice([0,0,1,4])
    job=plan_job(t_in,w_in,h,t_out,w_out,h,0)
    if job.wide: continue
    src=make_input(t_in,w_in,h,seed=it,valid=random.choice([True,False]),stride=sin,pad=0)
    first=random.randint(0,h-1); nr=random.randint(1,h-first)
    a=oracle_scale(src,t_in,w_in,h,t_out,w_out,h,0,sin,sout,first,nr)
    b,k=emu_scale(src,t_in,w_in,h,t_out,w_out,h,0,sin,sout,first,nr)
    seen[(k,job.filter_h,job.filter_v)]=seen.get((k,job.filter_h,job.filter_v),0)+1
    n+=1
    if not np.array_equal(a,b):
        bad+=1; print("MISMATCH",t_in,t_out,(w_in,h),(w_out,h),sin,sout,first,nr,k)
print("cases",n,"bad",bad,seen)
