"""Emulated k_up vs oracle: magnification on both axes, every type pair of the 64bpp class,
odd strides / misaligned rows, row slabs."""
import sys, random
sys.path.insert(0, "tests")
import numpy as np
from helpers import *
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 3); bad=0; n=0; seen={}
N=int(sys.argv[2]) if len(sys.argv) > 2 else 150
for it in range(N):
    w_in=random.choice([2,3,5,8,17,64,100,257]); h_in=random.choice([2,3,4,9,30,65])
    w_out=random.randint(w_in+1, min(w_in*9, 1300)); h_out=random.randint(h_in+1, min(h_in*9, 300))
    t_in=random.randrange(10); t_out=random.randrange(10)
    sin=w_in*bpp(t_in)+random.choice([0,0,1,3,4,16]); sout=w_out*bpp(t_out)+random.choice([0,0,1,4,16])
    src=make_input(t_in,w_in,h_in,seed=it,valid=random.choice([True,False]),stride=sin,pad=0)
    srgb=random.choice([0,1]) if len(sys.argv) > 4 else 0
    job=plan_job(t_in,w_in,h_in,t_out,w_out,h_out,srgb)
    if job.wide and not (len(sys.argv) > 3): continue
    first=random.randint(0,h_out-1); nr=random.randint(1,h_out-first)
    a=oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout,first,nr)
    b,k=emu_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout,first,nr)
    seen[(k,job.premul16,job.linear)]=seen.get((k,job.premul16,job.linear),0)+1
    n+=1
    if not np.array_equal(a,b):
        bad+=1; print("MISMATCH srgb",srgb,"types",t_in,t_out,(w_in,h_in),(w_out,h_out),sin,sout,first,nr,"class",k)
print("cases",n,"bad",bad,"classes",seen)
