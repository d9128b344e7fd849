import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import smolscale_amd as sm
W_IN=H_IN=4096; W_OUT=H_OUT=1024
job=sm.plan_job(0,W_IN,H_IN,0,W_OUT,H_OUT,0)
src=torch.randint(0,256,(8,H_IN,W_IN*4),dtype=torch.uint8,device="cuda")
dst=torch.zeros((8,H_OUT,W_OUT*4),dtype=torch.uint8,device="cuda")
ims=[]
for f in range(8):
    im=sm.SmolHipImage(); im.job=job; im.src=src[f].data_ptr(); im.dst=dst[f].data_ptr()
    im.rowstride_in, im.rowstride_out = W_IN*4, W_OUT*4; im.first_row, im.n_rows = 0, H_OUT; ims.append(im)
torch.cuda.synchronize()
b=sm.HipBatch(ims)
b.time(0,20); torch.cuda.synchronize()
for K in (1,2,5,10,50,100,500):
    best=1e9
    for rep in range(5):
        torch.cuda.synchronize(); t0=time.perf_counter(); ms=b.time(0,K); torch.cuda.synchronize(); t1=time.perf_counter()
        best=min(best,(t1-t0)*1e3)
    print("K=%d wall %.3f ms  events %.3f ms  per-step wall %.1f us events %.1f us  fixed %.0f us" % (K,best,ms,best*1e3/K,ms*1e3/K,(best-ms)*1e3))
