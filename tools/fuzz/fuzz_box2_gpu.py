"""GPU-only stress of the two-pass box path (host-pointer API), many small cases, to expose races."""
import sys, random
sys.path.insert(0,'tests')
import numpy as np
import helpers as H
rng=random.Random(int(sys.argv[1]) if len(sys.argv)>1 else 1); N=int(sys.argv[2]) if len(sys.argv)>2 else 2000
bad=0;n=0
for it in range(N):
    w_out=rng.choice([1,1,2,5,64,70]); h_out=rng.choice([1,1,2,5,9])
    w_in=min(int(w_out*rng.uniform(8.5,600))+1,4000); h_in=min(int(h_out*rng.uniform(8.5,400))+1,3000)
    t_in,t_out,srgb=rng.randrange(10),rng.randrange(10),rng.choice([0,1,1])
    job=H.plan_job(t_in,w_in,h_in,t_out,w_out,h_out,srgb)
    if job.filter_h!=3 or job.filter_v!=3: continue
    sin=w_in*H.bpp(t_in)+rng.choice([0,1,4]); sout=w_out*H.bpp(t_out)+rng.choice([0,3])
    src=H.make_input(t_in,w_in,h_in,seed=it,valid=bool(it&1),stride=sin,pad=0)
    a=H.oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout)
    b=H.hip_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout)
    n+=1
    rb=w_out*H.bpp(t_out)
    if not np.array_equal(H.rows_view(a,h_out,rb,sout),H.rows_view(b,h_out,rb,sout)):
        bad+=1
        b2=H.hip_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout)
        print("MISMATCH it",it,(t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout),"wide",job.wide,"retry ok" if np.array_equal(H.rows_view(a,h_out,rb,sout),H.rows_view(b2,h_out,rb,sout)) else "retry bad", flush=True)
print("cases",n,"bad",bad)
