#!/bin/bash
# The multi-GPU scaling curve in one command (a node with 2, 4 or 8 GPUs). One JSON line per N in
# {1, 2, 4, 8} (as far as the node goes) for
#   c2     4096^2 -> 1024^2, 32 frames per rank, output-row strips           (weak scaling: the driver's SCALE run)
#   c4     65535x4096 RGB8 -> 8192x4096, ONE image, rows sharded             (strong; kernel-only + the "e2e" host path)
#   c5mix  10 000 mixed-type thumbnails, image i on rank i % N               (strong)
# and the single-process host path over every GPU (SMOL_HIP_DEVICES=all: smol_scale_batch() cuts its rows into one
# strip per device, page-locked and pageable memory). Nothing here re-executes a process that has touched the GPU:
# each measurement is its own python process started from this shell.
#   bash tools/scale_curve.sh [outdir]      -> <outdir>/scale_curve.jsonl
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale}; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python3 -c "import torch; print(torch.cuda.device_count())")
: > "$OUT/scale_curve.jsonl"
port=29700
for cfg in c2 c4 c5mix; do
  for n in 1 2 4 8; do
    [ "$n" -gt "$NDEV" ] && continue
    port=$((port+1))
    if [ "$n" -eq 1 ]; then
      python3 bench.py --gpus 1 --steps 20 --warmup 3 --config $cfg --no-cpu-baseline --no-other-configs --no-e2e
    else
      python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
        bench.py --gpus $n --steps 20 --warmup 3 --config $cfg 2> "$OUT/err_${cfg}_$n.txt"
    fi | grep '^{' | tail -1 | tee -a "$OUT/scale_curve.jsonl" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$cfg N=$n value=%.1f %s ms_per_step=%.4f frac=%.4f ok=%s' % (d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['verified_vs_oracle']['all_ranks_ok']))"
  done
done
# one process, every device, host memory through the public API's backend entry (rows over devices)
for mem in pinned pageable; do
  python3 tools/bench_host_path.py --devices all --memory $mem 2> "$OUT/err_host_$mem.txt" | grep '^{' | tee -a "$OUT/scale_curve.jsonl"
done
echo "wrote $OUT/scale_curve.jsonl"
