"""Emulated k_linh with the vertical box sums in registers (JobDev::box2 == 3) + k_boxv vs oracle: a bilinear / COPY / ONE
horizontal axis under a TALL vertical box (ratio 64 and up): gamma-correct jobs, premul16 jobs and the class's compressed
storage (a box of 256+ rows without sRGB), every type pair, row ranges, any alignment."""
import sys, random
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from helpers import *
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 3); bad=0; n=0; seen={}
N=int(sys.argv[2]) if len(sys.argv) > 2 else 100
for it in range(N):
    w_out=random.choice([1,2,3,7,40,64,65,100,130,200,300]); h_out=random.choice([1,2,3,5,9])
    kh=random.choice(["down","up","copy","one","down","up"])
    w_in={"down":min(65535,max(w_out+1,int(w_out*random.uniform(1.01,7.99)))),"up":max(2,int(w_out/random.uniform(1.05,9))),"copy":w_out,"one":1}[kh]
    if kh=="one" and w_out==1: continue
    if kh in ("down","up") and w_in==w_out: continue
    ratio=random.choice([64,65,100,129,255,256,257,300,500,1000,2500])
    h_in=min(65535,int(h_out*ratio*random.uniform(1.0,1.3))+random.randint(0,3))
    if w_in*h_in>6_000_000: continue
    mode=random.choice(["srgb","srgb","p16","flat"])
    if mode=="srgb": srgb=1; t_in=random.randrange(10); t_out=random.randrange(10)
    elif mode=="p16": srgb=0; t_in=random.choice([4,5,6,7]); t_out=random.choice([4,5,6,7])
    else: srgb=0; t_in=random.choice([0,1,2,3,8,9]); t_out=random.choice([0,1,2,3,8,9])
    sin=w_in*bpp(t_in)+random.choice([0,0,4,16,1,3]); sout=w_out*bpp(t_out)+random.choice([0,0,1,4,16])
    src=make_input(t_in,w_in,h_in,seed=it,valid=random.choice([True,False]),stride=sin,pad=0)
    job=plan_job(t_in,w_in,h_in,t_out,w_out,h_out,srgb)
    if not job.wide: continue
    first=random.randint(0,h_out-1); nr=random.randint(1,h_out-first)
    a=oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout,first,nr)
    b,k=emu_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout,first,nr)
    name=product().smol_hip_kernel_name
    import ctypes
    name.restype=ctypes.c_char_p
    kn=name(ctypes.byref(job), ctypes.c_void_p(16), sin, ctypes.c_void_p(16), sout, first, nr).decode()
    key=(kh,mode,kn[:28])
    seen[key]=seen.get(key,0)+1
    n+=1
    if not np.array_equal(a,b):
        bad+=1; print("MISMATCH",kh,mode,"types",t_in,t_out,(w_in,h_in),(w_out,h_out),sin,sout,first,nr,"class",k,kn)
print("cases",n,"bad",bad)
for k,v in sorted(seen.items()): print("  ",k,v)
