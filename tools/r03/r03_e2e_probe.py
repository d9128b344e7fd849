import sys, time, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import smolscale_amd as sm
W,Hh,w,h=4096,4096,1024,1024
rng=np.random.default_rng(1)
res={}
for mode in ("pinned","pageable"):
    if mode=="pinned":
        a,b=sm.PinnedArray(W*Hh*4),sm.PinnedArray(w*h*4); src,dst=a.array,b.array; src[:]=rng.integers(0,256,W*Hh*4,dtype=np.uint8)
    else:
        src=rng.integers(0,256,W*Hh*4,dtype=np.uint8); dst=np.zeros(w*h*4,np.uint8)
    def ctx_call():
        with sm.SmolScaleCtx(src,0,W,Hh,W*4,dst,0,w,h,w*4) as c: c.batch(0,h)
    ctx_call(); 
    t=[]; 
    for _ in range(5):
        t0=time.perf_counter(); ctx_call(); t.append(time.perf_counter()-t0)
    res[mode+"_ctx_per_frame_ms"]=round(min(t)*1e3,3)
    c=sm.SmolScaleCtx(src,0,W,Hh,W*4,dst,0,w,h,w*4); c.batch(0,h)
    t=[]
    for _ in range(5):
        t0=time.perf_counter(); c.batch(0,h); t.append(time.perf_counter()-t0)
    res[mode+"_ctx_kept_ms"]=round(min(t)*1e3,3)
    # in 4 row batches
    t=[]
    for _ in range(5):
        t0=time.perf_counter()
        for q in range(4): c.batch(q*256,256)
        t.append(time.perf_counter()-t0)
    res[mode+"_ctx_kept_4_batches_ms"]=round(min(t)*1e3,3)
    c.destroy()
print(json.dumps(res))
