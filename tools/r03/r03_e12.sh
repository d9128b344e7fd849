export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
export SMOL_VEC16_ANY=1
run() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
CF=R1.03,R1.1,R1.2,R1.35,R1.5,R1.7,R1.9,HDOWNVUP
run $CF cpl1
for r in 4 6 8 10 12 16; do SMOL_TILE_CPL=4 SMOL_TILE_ROWS=$r run $CF cpl4.rows$r; done
SMOL_TILE_CPL=4 run $CF cpl4.rowsdefault
