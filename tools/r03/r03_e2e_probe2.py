import sys, time, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import smolscale_amd as sm
W,Hh,w,h=4096,4096,1024,1024
rng=np.random.default_rng(1)
res={}
def timeit(src,dst,n=5):
    def ctx_call():
        with sm.SmolScaleCtx(src,0,W,Hh,W*4,dst,0,w,h,w*4) as c: c.batch(0,h)
    ctx_call(); t=[]
    for _ in range(n):
        t0=time.perf_counter(); ctx_call(); t.append(time.perf_counter()-t0)
    return round(min(t)*1e3,3)
a,b=sm.PinnedArray(W*Hh*4),sm.PinnedArray(w*h*4); a.array[:]=rng.integers(0,256,W*Hh*4,dtype=np.uint8)
res["one_pinned_pair"]=timeit(a.array,b.array)
extra=[sm.PinnedArray(W*Hh*4) for _ in range(15)]
res["with_15_more_registered"]=timeit(a.array,b.array)
print(json.dumps(res))
