export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
mkdir -p gpurun_out/${TAG:-r02d}
for cfg in ${CFGS:-BOX R1.5 R3 R6 PHOTO PNG}; do
  for rows in ${ROWS:-default 1 2 3 4 6 8 16}; do
    for band in 1; do
      if [ $rows = default ]; then unset SMOL_TILE_ROWS; else export SMOL_TILE_ROWS=$rows; fi
      python tools/bench_configs.py --configs $cfg --iters 20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$cfg rows=$rows', d['kernel'], d['step_us'], d['step_us_min'], d['frac_of_8TBps'])
"
    done
  done
done | tee gpurun_out/${TAG:-r02d}/tile_rows_sweep.txt
