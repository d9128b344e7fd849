export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
CF2=C2,C2CROP0,C2CROP4,C2CROP12,C2ODD
for rep in 1 2; do
run $CF2 pow2.swz
SMOL_POW2_NO_SWZ=1 run $CF2 pow2.noswz
SMOL_POW2_ALIGN16=1 run $CF2 tile
done
for by in 1 2 4 8; do SMOL_POW2_BY=$by run $CF2 pow2.swz.by=$by; done
