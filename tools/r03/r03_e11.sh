export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
CF=R1.03,R1.1,R1.2,R1.35,R1.5,U1.1,U1.37,U2.6,U3
run $CF base
SMOL_VEC16_ANY=1 run $CF vec16any
SMOL_TILE_CPL=4 run $CF cpl4
SMOL_TILE_CPL=4 SMOL_VEC16_ANY=1 run $CF cpl4.vec16any
SMOL_TILE_CPL=4 SMOL_VEC16_ANY=1 SMOL_TILE_ROWS=8 run $CF cpl4.vec16any.rows8
for r in 4 6 8 12 16; do SMOL_VEC16_ANY=1 SMOL_TILE_ROWS=$r run U1.1,U1.37 up.rows$r; done
