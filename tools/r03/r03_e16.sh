export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
CF=PHOTO,PHOTOODD,PHOTO32ODD,THUMB,C4
for rep in 1 2; do
SMOL_OUT3_ALIGNED=1 run $CF old
run $CF new
done
