export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
CF=S4K97,S4K89,S4K81,S4K73,S4K65,S4K57,S8K93,S4K110,S4K140
run $CF default
for r in 2 3 4 6 8 12 16; do SMOL_TILE_ROWS=$r run $CF rows$r; done
SMOL_TILE_CPL=1 run $CF cpl1
