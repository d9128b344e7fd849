import sys, random
sys.path.insert(0, "tests")
import numpy as np
from helpers import *
random.seed(int(sys.argv[1])); N=int(sys.argv[2])
def dims():
    cls = random.choice(["box","box","boxbig","copy","0h","1h","up","one"])
    if cls=="box":
        o=random.randint(1,80); return random.randint(8*o+1, min(24*o, 4000)), o
    if cls=="boxbig":
        o=random.randint(1,12); return random.randint(24*o+1, min(255*o, 3000)), o
    if cls=="copy":
        d=random.randint(1,100); return d,d
    if cls=="one": return 1, random.randint(1,80)
    if cls=="up":
        i=random.randint(2,40); return i, random.randint(i+1,4*i)
    if cls=="0h":
        o=random.randint(1,80); return random.randint(o+1,2*o), o
    o=random.randint(1,60); return random.randint(2*o+1,4*o), o
bad=0; kc={}; it=0; done=0
while done<N:
    it+=1
    w_in,w_out=dims(); h_in,h_out=dims()
    if w_in*h_in>600000 or w_out*h_out>8000: continue
    while True:
        t_in=random.randrange(10); t_out=random.randrange(10)
        if not (4<=t_in<=7 and 4<=t_out<=7): break
    job=plan_job(t_in,w_in,h_in,t_out,w_out,h_out,0)
    if 3 not in (job.filter_h, job.filter_v) or job.wide: continue
    done+=1
    sin=w_in*bpp(t_in)+random.choice([0,0,1,3,4,16]); sout=w_out*bpp(t_out)+random.choice([0,0,1,5,8])
    mis=random.choice([0,0,1,2,4])
    raw=make_input(t_in,w_in,h_in,seed=it,valid=random.random()<.5,stride=sin,pad=32)
    buf=np.zeros(raw.size+32,np.uint8); off=(-buf.ctypes.data)%16+mis; src=buf[off:off+raw.size]; src[:]=raw
    first=random.randint(0,h_out-1); n=random.randint(1,h_out-first)
    a=oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,0,sin,sout,first,n)
    b,k=emu_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,0,sin,sout,first,n)
    kc[k]=kc.get(k,0)+1
    if not np.array_equal(a,b):
        bad+=1
        rb=w_out*bpp(t_out); A=rows_view(a,n,rb,sout); B=rows_view(b,n,rb,sout); nz=np.argwhere(A!=B)
        print("MISMATCH",TYPE_NAMES[t_in],(w_in,h_in),"->",TYPE_NAMES[t_out],(w_out,h_out),"rows",first,n,"filt",job.filter_h,job.filter_v,"halv",job.halvings_h,job.halvings_v,"k",k,"ndiff",len(nz),nz[:3].tolist() if len(nz) else "PAD", A[tuple(nz[0])] if len(nz) else "", B[tuple(nz[0])] if len(nz) else "")
        if bad>8: break
print("cases",done,"bad",bad,kc)
