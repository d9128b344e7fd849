export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
mkdir -p gpurun_out/r02e
for cfg in C3 P16UP; do
  for band in band noband; do
  for rows in default 2 4 8 16; do
      if [ $rows = default ]; then unset SMOL_UP_ROWS; else export SMOL_UP_ROWS=$rows; fi
      if [ $band = band ]; then unset SMOL_NO_UP_BAND; else export SMOL_NO_UP_BAND=1; fi
      python tools/bench_configs.py --configs $cfg --iters 20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$cfg $band rows=$rows', d['kernel'], d['step_us'], d['step_us_min'], d['frac_of_8TBps'])
"
  done
  done
done | tee gpurun_out/r02e/up_sweep.txt
