export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
for rows in default 4 8 12 16 24; do
  if [ $rows = default ]; then unset SMOL_UP_ROWS; else export SMOL_UP_ROWS=$rows; fi
  python tools/bench_configs.py --configs SRGBUP,SRGBPNGUP,P16UP --iters 20 2>/dev/null | python -c "
import sys,json
print('rows=$rows', ' '.join('%s=%.1f(%.3f)' % (d['config'], d['step_us'], d['frac_of_8TBps']) for d in map(json.loads, sys.stdin)))"
done
unset SMOL_UP_ROWS
python tools/bench_configs.py --configs SRGBPHOTO,SRGBHALF,SRGBPNG,P16HALF --iters 20 2>/dev/null | python -c "
import sys,json
print('k_down new default', ' '.join('%s=%.1f(%.3f)' % (d['config'], d['step_us'], d['frac_of_8TBps']) for d in map(json.loads, sys.stdin)))"
