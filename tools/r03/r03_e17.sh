export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
CF=S4K49,R2.5,R3,R3.5,S4K25,R5,R6,R7
run $CF cpl1
for r in 2 3 4 6; do SMOL_TILE_CPL=4 SMOL_TILE_ROWS=$r run $CF cpl4.rows$r; done
