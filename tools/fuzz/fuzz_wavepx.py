import sys, random
sys.path.insert(0, "tests")
import numpy as np
from helpers import *
random.seed(3); bad=0; kc={}
for it in range(60):
    h_out=random.randint(1,5); h_in=random.randint(32*h_out+1, min(65535, 400*h_out))
    w_in,w_out=random.choice([(random.randint(1,40),)*2, (random.randint(300,900), random.randint(1,3)), (random.randint(20,60), random.randint(3,9)), (1, random.randint(1,9))])
    t_in=random.randrange(10); t_out=random.randrange(10); srgb=random.randrange(2)
    job=plan_job(t_in,w_in,h_in,t_out,w_out,h_out,srgb)
    if not job.wide: continue
    src=make_input(t_in,w_in,h_in,seed=it,valid=bool(it&1),pad=16)
    first=random.randint(0,h_out-1); n=random.randint(1,h_out-first)
    a=oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,first=first,n=n)
    b,k=emu_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,first=first,n=n)
    kc[k]=kc.get(k,0)+1
    if not np.array_equal(a,b):
        bad+=1; print("MISMATCH",t_in,(w_in,h_in),t_out,(w_out,h_out),srgb,first,n,k)
print("bad",bad,kc)
