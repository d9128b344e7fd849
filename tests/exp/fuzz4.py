import sys, random
sys.path.insert(0, "tests")
import numpy as np
from helpers import *
random.seed(int(sys.argv[1])); N=int(sys.argv[2])
def dims():
    cls = random.choice(["one","copy","up","0h","1h","2h","tiny","wideup","widedown"])
    if cls=="one": return 1, random.randint(1,200)
    if cls=="copy":
        d=random.randint(1,300); return d,d
    if cls=="up":
        i=random.randint(2,60); return i, random.randint(i+1,6*i+3)
    if cls=="0h":
        o=random.randint(1,150); return random.randint(o+1,2*o), o
    if cls=="1h":
        o=random.randint(1,100); return random.randint(2*o+1,4*o), o
    if cls=="2h":
        o=random.randint(1,70); return random.randint(4*o+1,8*o), o
    if cls=="wideup":
        i=random.randint(40,120); return i, random.randint(300,700)
    if cls=="widedown":
        o=random.randint(65,200); return random.randint(o+1,8*o), o
    return random.randint(1,4), random.randint(1,4)
bad=0; kc={}
for it in range(N):
    w_in,w_out=dims(); h_in,h_out=dims()
    if h_out>80: h_out=random.randint(1,80); h_in=max(1,min(h_in,8*h_out)); 
    while True:
        t_in=random.randrange(10); t_out=random.randrange(10)
        if not (4<=t_in<=7 and 4<=t_out<=7): break
    sin=w_in*bpp(t_in)+random.choice([0,0,1,3,4,7,16]); sout=w_out*bpp(t_out)+random.choice([0,0,1,2,5,8])
    mis=random.choice([0,0,1,2,3,4,8])
    raw=make_input(t_in,w_in,h_in,seed=it,valid=random.random()<.5,stride=sin,pad=32)
    buf=np.zeros(raw.size+32,np.uint8); off=(-buf.ctypes.data)%16+mis; src=buf[off:off+raw.size]; src[:]=raw
    first=random.randint(0,h_out-1); n=random.randint(1,h_out-first)
    a=oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,0,sin,sout,first,n)
    b,k=emu_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,0,sin,sout,first,n,force_general=2)
    kc[k]=kc.get(k,0)+1
    if not np.array_equal(a,b):
        bad+=1
        rb=w_out*bpp(t_out); A=rows_view(a,n,rb,sout); B=rows_view(b,n,rb,sout); nz=np.argwhere(A!=B)
        job=plan_job(t_in,w_in,h_in,t_out,w_out,h_out,0)
        print("MISMATCH",TYPE_NAMES[t_in],(w_in,h_in),"->",TYPE_NAMES[t_out],(w_out,h_out),"rows",first,n,"sin",sin,"mis",mis,"filt",job.filter_h,job.filter_v,"halv",job.halvings_h,job.halvings_v,"k",k,"ndiff",len(nz),nz[:3].tolist() if len(nz) else "PAD")
        if bad>8: break
print("cases",it+1,"bad",bad,kc)
