"""Directed GPU cases whose workgroups need more than 64 KB of dynamic LDS."""
import sys
sys.path.insert(0,'tests')
import numpy as np
import helpers as H
import smolscale_amd as sm
cases=[(4,2000,100,4,2050,110,1,"k_upg<32> mild magnification (73 KB)"),
       (0,4000,600,0,64,10,0,"k_boxh narrow ratio 62 (123 KB)"),
       (0,4000,600,0,64,10,1,"k_boxh wide ratio 62 (130 KB)"),
       (8,4000,100,8,3990,99,1,"k_downg mild minification"),
       (0,5000,40,0,4990,41,0,"mixed axes")]
for (t_in,w_in,h_in,t_out,w_out,h_out,srgb,what) in cases:
    src=H.make_input(t_in,w_in,h_in,seed=1,valid=True,pad=0)
    want=H.oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb)
    try:
        got=H.hip_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb)
        print(what, "OK" if np.array_equal(got,want) else "MISMATCH")
    except Exception as e:
        print(what, "EXCEPTION", e)
