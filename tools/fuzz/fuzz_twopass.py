"""Emulated two-pass road of the 128bpp class (k_linh / k_boxh + k_linv / k_boxv) vs oracle: axes scaled independently --
one minified and one magnified, a COPY axis, bilinear x box -- gamma-correct jobs of every type pair and premul16 jobs
(unassociated in and out) with and without sRGB."""
import sys, random
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from helpers import *
import smolscale_amd as sm
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 3); bad=0; n=0; seen={}
N=int(sys.argv[2]) if len(sys.argv) > 2 else 150
def axis(kind, d_out):
    if kind == "one": return 1
    if kind == "copy": return d_out
    if kind == "up": return max(2, int(d_out / random.uniform(1.05, 9)))
    if kind == "down": return min(65535, max(d_out + 1, int(d_out * random.uniform(1.01, 7.99))))
    return min(65535, int(d_out * random.uniform(8.5, 40)) + 1)          # box
for it in range(N):
    w_out=random.choice([1,2,3,7,40,64,65,130,256,300]); h_out=random.choice([1,2,5,17,33,70])
    kh, kv = random.choice([("down","up"),("up","down"),("copy","down"),("down","copy"),("copy","up"),("up","copy"),
                            ("box","down"),("box","up"),("box","copy"),("down","box"),("up","box"),("copy","box"),("copy","copy"),
                            ("one","down"),("one","up"),("down","one"),("up","one"),("one","box"),("box","one"),("one","copy"),("copy","one")])
    w_in, h_in = axis(kh, w_out), axis(kv, h_out)
    if (kh != "copy" and w_in == w_out) or (kv != "copy" and h_in == h_out): continue
    if (w_in < 2 and kh != "one") or (h_in < 2 and kv != "one"): continue
    if (kh == "one" and w_out == 1) or (kv == "one" and h_out == 1): continue
    srgb=random.choice([1,1,0])
    if srgb: t_in=random.randrange(10); t_out=random.randrange(10)
    else: t_in=random.choice([4,5,6,7]); t_out=random.choice([4,5,6,7])     # premul16: the 128bpp class without sRGB
    sin=w_in*bpp(t_in)+random.choice([0,0,4,16,1,3]); sout=w_out*bpp(t_out)+random.choice([0,0,1,4,16])
    src=make_input(t_in,w_in,h_in,seed=it,valid=random.choice([True,False]),stride=sin,pad=0)
    job=plan_job(t_in,w_in,h_in,t_out,w_out,h_out,srgb)
    if not job.wide: continue
    first=random.randint(0,h_out-1); nr=random.randint(1,h_out-first)
    a=oracle_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout,first,nr)
    b,k=emu_scale(src,t_in,w_in,h_in,t_out,w_out,h_out,srgb,sin,sout,first,nr)
    key=(kh,kv,k,job.premul16,srgb)
    seen[key]=seen.get(key,0)+1
    n+=1
    if not np.array_equal(a,b):
        bad+=1; print("MISMATCH",kh,kv,"types",t_in,t_out,"srgb",srgb,(w_in,h_in),(w_out,h_out),sin,sout,first,nr,"class",k,"p16",job.premul16,"halv",job.halvings_h,job.halvings_v)
print("cases",n,"bad",bad)
tp = sum(v for k,v in seen.items() if k[2]==10); print("two-pass (class 10):", tp, " p16:", sum(v for k,v in seen.items() if k[2]==10 and k[3]))
for k,v in sorted(seen.items()):
    if k[2] != 10: print("  not two-pass:",k,v)
