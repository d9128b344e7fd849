import sys, random
sys.path.insert(0, "tests")
import numpy as np
from helpers import *
random.seed(11); bad=0; n=0
for it in range(80):
    w_out=random.choice([64,128,192,256,320,512,68,260,300]); h_out=random.randint(1,20)
    mode=random.choice(["up","copy","0h","1h","2h"])
    w_in={"up":random.randint(20,w_out-1),"copy":w_out,"0h":random.randint(w_out+1,2*w_out),"1h":random.randint(2*w_out+1,4*w_out),"2h":random.randint(4*w_out+1,8*w_out)}[mode]
    h_in=random.choice([h_out, random.randint(2,50), random.randint(h_out+1,4*h_out)])
    t_in=random.randrange(10); t_out=random.choice([8,9])
    sout=w_out*3+random.choice([0,0,4,8]) ; sout += (-sout)%4
    src=make_input(t_in,w_in,h_in,seed=it,valid=True,pad=32)
    job=plan_job(t_in,w_in,h_in,t_out,w_out,h_out,0)
    if job.wide or 3 in (job.filter_h,job.filter_v): continue
    first=random.randint(0,h_out-1); nr=random.randint(1,h_out-first)
    a=oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,0,None,sout,first,nr)
    b,k=emu_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,0,None,sout,first,nr,force_general=2)
    n+=1
    if not np.array_equal(a,b):
        bad+=1; print("MISMATCH",mode,t_in,(w_in,h_in),t_out,(w_out,h_out),sout,first,nr)
print("cases",n,"bad",bad)
