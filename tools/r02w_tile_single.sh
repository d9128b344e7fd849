# k_tile: one staging buffer without prefetch (SMOL_TILE_SINGLE) x staging budget, tuning build
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
for single in 0 1; do for b in 2048 4096 8192; do
env $( [ $single = 1 ] && echo SMOL_TILE_SINGLE=1 ) SMOL_TILE_LDS_BUDGET=$b python tests/bench_configs_checked.py --configs R1.5,R3,R6,PHOTO,PNG,BOX --iters 20 2>/dev/null | python -c "
import sys,json
print('single=$single budget=$b', ' '.join('%s=%.1f%s' % (d['config'], d['step_us'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"
done; done
