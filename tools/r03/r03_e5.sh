export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tools/bench_configs.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)' % (d['config'], d['step_us'], d['frac_of_8TBps']) for d in map(json.loads, sys.stdin)))"; }
CF=BOX,R5,R3
for r in 1 2 4 7; do for d in 0 5; do SMOL_TILE_ROWS=$r SMOL_TILE_DBG=$d run $CF rows=$r.dbg=$d; done; done
for b in 2400 4800 7200; do for d in 0 5; do SMOL_TILE_ROWS=2 SMOL_TILE_LDS_BUDGET=$b SMOL_TILE_DBG=$d run $CF rows=2.budget=$b.dbg=$d; done; done
