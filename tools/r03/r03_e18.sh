export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
CF=THUMB,S1080T
run $CF box2
for r in 1 2 3; do for b in 4096 8192; do SMOL_BOX2_MIN=30 SMOL_TILE_ROWS=$r SMOL_TILE_LDS_BUDGET=$b run $CF tile.rows$r.budget$b; done; done
