"""GPU parity with with_srgb = 1 and mostly-opaque alpha (fused-table fast path), all kernel classes."""
import sys, random
sys.path.insert(0, "tests")
import numpy as np
import helpers as H
rng=random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 3); bad=0; n=0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2000):
    w_in=rng.choice([2,5,16,64,100,257,1000]); h_in=rng.choice([2,4,9,33,200])
    w_out=rng.choice([1,3,8,32,64,90,300,500,2000]); h_out=rng.choice([1,2,5,16,40,100,400])
    if rng.random() < 0.3:
        h=rng.randrange(3); w_in, h_in = w_out << (h+1), h_out << (h+1)
        if w_in*h_in > 2_000_000: continue
    t_in=rng.randrange(8); t_out=rng.randrange(10)
    src=H.make_input(t_in,w_in,h_in,seed=it,valid=True,pad=0)
    apos=3 if (t_in & 3) < 2 else 0
    a=src[apos::4]
    a[np.random.default_rng(it).random(a.size) < 0.9]=255
    want=H.oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,1)
    got=H.hip_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,1)
    n+=1
    if not np.array_equal(want,got):
        bad+=1; print("MISMATCH",t_in,t_out,(w_in,h_in),(w_out,h_out), flush=True)
print("cases",n,"bad",bad)
