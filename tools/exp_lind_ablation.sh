#!/bin/bash
# Where k_lind's time goes, by ablation (needs smolscale_amd/libsmolscale_amd_tuning.so: tools/build_tuning.sh).
# SMOL_LIND_DBG bits: 1 no taps, 2 no loads, 4 no stores, 8 source look-ups -> shifts, 16 output look-ups -> shifts, 32 no conversion.
# usage (through gpurun): bash tools/exp_lind_ablation.sh > gpurun_out/lind_ablation.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f' % (d['config'].split(':')[0][:10], d['step_us']) for d in map(json.loads, sys.stdin)))"; }
C=SRGBHALF,SRGBU2,SRGBR6,SRGBVCOPY,SRGBPNG,P16HALF
for dbg in 0 1 2 4 8 16 24 32 33 6 7 39 63 0; do
  echo "dbg=$dbg  $(SMOL_LIND_DBG=$dbg f $C)"
done
