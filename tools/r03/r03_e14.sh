export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
CF=S8K93,S4K97,S4K176,S4K193,S1080T,S4K49,S4K25,PHOTO,PHOTOODD,PHOTO32ODD
run $CF base
SMOL_NO_UP=1 run S4K176,S4K193 noup
SMOL_TILE_ROWS=4 run S8K93,S4K97 rows4
SMOL_TILE_ROWS=12 run S8K93,S4K97 rows12
SMOL_NO_TILE_SWZ=1 SMOL_TILE_NO_SWZ=1 run S8K93,S4K97,S4K49,S4K25 noswz
