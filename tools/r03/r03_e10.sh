export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
CF=R1.03,R1.1,R1.2,R1.35,U1.1,U1.3
run $CF base
SMOL_TILE_CPL=4 run $CF cpl4
SMOL_TILE_WG1=1 run $CF wg1
SMOL_TILE_NO_SINGLE=1 run $CF nosingle
SMOL_TILE_CPL=4 SMOL_TILE_ROWS=8 run $CF cpl4.rows8
SMOL_TILE_ROWS=8 run $CF rows8
SMOL_TILE_ROWS=32 run $CF rows32
SMOL_TILE_DBG=4 run $CF nostore
SMOL_TILE_DBG=5 run $CF stageonly
SMOL_TILE_DBG=2 run $CF nostage
