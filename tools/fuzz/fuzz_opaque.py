"""Emulated kernels vs oracle with with_srgb = 1 and mostly-opaque alpha (the fused-table fast paths)."""
import sys, random
sys.path.insert(0, "tests")
import numpy as np
from helpers import *
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 3); bad=0; n=0; seen={}
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    w_in=random.choice([2,5,16,64,100,257]); h_in=random.choice([2,4,9,33])
    w_out=random.choice([1,3,8,32,64,90,300]); h_out=random.choice([1,2,5,16,40])
    t_in=random.randrange(8); t_out=random.randrange(10)
    src=make_input(t_in,w_in,h_in,seed=it,valid=True,pad=0)
    # alpha byte: 3 for RGBA/BGRA (types 0,1,4,5), 0 for ARGB/ABGR; 90 % of the pixels opaque
    apos=3 if (t_in & 3) < 2 else 0
    a=src[apos::4]
    mask=np.random.default_rng(it).random(a.size) < 0.9
    if t_in < 4:
        px=src.reshape(-1,4); keep=~mask
        # premultiplied input: an opaque pixel may hold any colour
        a[mask]=255
    else:
        a[mask]=255
    job=plan_job(t_in,w_in,h_in,t_out,w_out,h_out,1)
    want=oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,1)
    got,k=emu_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,1)
    seen[k]=seen.get(k,0)+1; n+=1
    if not np.array_equal(want,got):
        bad+=1; print("MISMATCH",t_in,t_out,(w_in,h_in),(w_out,h_out),"class",k)
print("cases",n,"bad",bad,seen)
