import sys
sys.path.insert(0,'tests')
import numpy as np
import helpers as H
t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout,first,n=(5, 90, 409, 8, 1, 1, 1, 364, 11, 0, 1)
bad=0
for seed in range(300):
    src=H.make_input(t_in,w_in,h_in,seed=seed,valid=bool(seed&1),stride=sin,pad=0)
    a=H.oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout,first,n)
    for rep in range(2):
        b=H.hip_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout,first,n)
        if not np.array_equal(a[:3],b[:3]):
            bad+=1; print("seed",seed,"rep",rep,a[:3],b[:3])
print("bad",bad)
