export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
mkdir -p gpurun_out/r02e
for rep in 1 2; do
for cfg in C3; do
  for rows in 5 6 7 8 9 10 12 14; do
      export SMOL_UP_ROWS=$rows
      python tools/bench_configs.py --configs $cfg --iters 20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$cfg rows=$rows', d['kernel'], d['step_us'], d['step_us_min'], d['frac_of_8TBps'])
"
  done
done
done | tee gpurun_out/r02e/up_sweep2.txt
