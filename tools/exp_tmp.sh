cd "$GRAFT_REPO_ROOT" || exit 1
p() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); e=d['e2e']
print(sys.argv[2], {k:(v['ms'],v['GBps_over_pcie']) for k,v in e.items() if isinstance(v,dict)})" $1 "$2"; }
python3 bench.py --config c4 --steps 5 --warmup 2 > gpurun_out/c4_a.json 2>/dev/null; p gpurun_out/c4_a.json "c4 mapped 4MB"
SMOL_HIP_MAPPED_CHUNK_KB=8192 python3 bench.py --config c4 --steps 5 --warmup 2 > gpurun_out/c4_b.json 2>/dev/null; p gpurun_out/c4_b.json "c4 mapped 8MB"
SMOL_HIP_MAPPED=0 python3 bench.py --config c4 --steps 5 --warmup 2 > gpurun_out/c4_c.json 2>/dev/null; p gpurun_out/c4_c.json "c4 DMA"
python3 bench.py --config c4 --steps 5 --warmup 2 > gpurun_out/c4_a.json 2>/dev/null; p gpurun_out/c4_a.json "c4 mapped 4MB"
SMOL_HIP_MAPPED=0 python3 bench.py --config c4 --steps 5 --warmup 2 > gpurun_out/c4_c.json 2>/dev/null; p gpurun_out/c4_c.json "c4 DMA"
for g in "8 65535 512 0 8192 512" "8 65535 1024 0 8192 1024" "8 65535 2048 0 8192 2048" "0 8192 8192 0 2048 2048" "0 16384 8192 0 4096 2048"; do
 for m in 1 0; do
  echo "MAPPED=$m $g: $(SMOL_HIP_MAPPED=$m python3 - $g <<'PY'
import sys
sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import bench_rows_per_call as B
g=tuple(int(x) for x in sys.argv[1:7])
r=B.measure(g, False, 0, reps=4)
mb=(g[1]*g[2]*(3 if g[0]>=8 else 4)+g[4]*g[5]*(3 if g[3]>=8 else 4))/1e6
print("%.3f ms %.1f GB/s ok=%s"%(r['ms_per_image'], mb/r['ms_per_image'], r['equal_to_whole_image_call']))
PY
)"
 done
done
