import sys, time, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
torch.cuda.set_device(0); x=torch.zeros(1<<30,dtype=torch.uint8,device="cuda"); torch.cuda.synchronize()
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
res["pinned_in_place_with_torch_loaded"]=timeit(a.array,b.array)
ts=torch.empty(W*Hh*4,dtype=torch.uint8,pin_memory=True); td=torch.empty(w*h*4,dtype=torch.uint8,pin_memory=True)
res["torch_pin_memory_with_torch_loaded"]=timeit(ts.numpy(),td.numpy())
src=rng.integers(0,256,W*Hh*4,dtype=np.uint8); dst=np.zeros(w*h*4,np.uint8)
res["pageable_with_torch_loaded"]=timeit(src,dst)
print(json.dumps(res))
