"""Emulated k_pow2x vs oracle: every type pair, srgb on/off, 2:1/4:1/8:1, thumbnails and wider rows."""
import sys, random
sys.path.insert(0, "tests")
import numpy as np
from helpers import *
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 5); bad=0; n=0; seen={}
N=int(sys.argv[2]) if len(sys.argv) > 2 else 150
for it in range(N):
    H=random.randrange(3); R=2<<H
    w_out=random.choice([8,16,24,32,40,48,64,72,96,128,136,200,264]); h_out=random.randint(1,40)
    w_in=w_out*R; h_in=h_out*R
    t_in=random.randrange(10); t_out=random.randrange(10); srgb=random.choice([0,0,1])
    src=make_input(t_in,w_in,h_in,seed=it,valid=random.choice([True,False]),pad=32)
    first=random.randint(0,h_out-1); nr=random.randint(1,h_out-first)
    a=oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,None,None,first,nr)
    b,k=emu_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,None,None,first,nr)
    seen[k]=seen.get(k,0)+1
    n+=1
    if not np.array_equal(a,b):
        bad+=1; print("MISMATCH H",H,"types",t_in,t_out,"srgb",srgb,(w_in,h_in),(w_out,h_out),first,nr,"class",k)
print("cases",n,"bad",bad,"classes",seen)
