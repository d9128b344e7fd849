"""Emulated k_lind, 24-bit sources (the three-channel flavour lind3_body) vs oracle: gamma-correct jobs from RGB8 / BGR8 rows into
every output type -- minified, magnified and mixed axes, COPY axes, any row alignment, row ranges."""
import sys, random
sys.path.insert(0, "tests")
import numpy as np
from helpers import *
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 3); bad=0; n=0; seen={}
N=int(sys.argv[2]) if len(sys.argv) > 2 else 150
for it in range(N):
    w_out=random.choice([1,2,3,7,40,64,65,127,130,256,300,515]); h_out=random.choice([1,2,5,17,33,70])
    mode=random.choice(["down","down","up","mixed","copyx","copyy"])
    def dim(o, kind):
        if kind=="down": return max(2,min(int(o*random.uniform(1.01,7.99))+random.randint(0,1), 8*o-1))
        if kind=="up": return max(2,int(o/random.uniform(1.05,6.0)))
        return o
    kx = {"down":"down","up":"up","mixed":random.choice(["down","up"]),"copyx":"copy","copyy":random.choice(["down","up"])}[mode]
    ky = {"down":"down","up":"up","mixed":"up" if kx=="down" else "down","copyx":random.choice(["down","up"]),"copyy":"copy"}[mode]
    w_in=dim(w_out,kx); h_in=dim(h_out,ky)
    t_in=random.choice([8,9]); t_out=random.randrange(10)
    sin=w_in*3+random.choice([0,0,4,16,1,3]); sout=w_out*bpp(t_out)+random.choice([0,0,1,4,16])
    src=make_input(t_in,w_in,h_in,seed=it,valid=True,stride=sin,pad=0)
    job=plan_job(t_in,w_in,h_in,t_out,w_out,h_out,1)
    if not job.wide: continue
    first=random.randint(0,h_out-1); nr=random.randint(1,h_out-first)
    a=oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,1,sin,sout,first,nr)
    b,k=emu_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,1,sin,sout,first,nr)
    key=(k,job.halvings_h,job.halvings_v,job.filter_h,job.filter_v)
    seen[key]=seen.get(key,0)+1
    n+=1
    if not np.array_equal(a,b):
        bad+=1; print("MISMATCH types",t_in,t_out,(w_in,h_in),(w_out,h_out),sin,sout,first,nr,"class",k,"halv",job.halvings_h,job.halvings_v,"filters",job.filter_h,job.filter_v)
print("cases",n,"bad",bad,"(class,hh,hv,fh,fv):",sorted(seen.items()))
