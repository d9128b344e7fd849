export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
CF=U1.1,U1.3,U1.5,U1.75,U2,U3
run $CF k_up
SMOL_NO_UP=1 run $CF k_tile
for r in 4 6 12 16; do SMOL_UP_ROWS=$r run $CF k_up.rows=$r; done
