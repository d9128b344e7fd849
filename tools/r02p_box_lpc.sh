# k_tile box: lanes per column threshold (SMOL_TILE_BOX_LPC8_MIN) and tile rows, tuning build
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
for m in 24 12 8 4; do for rows in default 2 4; do
if [ $rows = default ]; then unset SMOL_TILE_ROWS; else export SMOL_TILE_ROWS=$rows; fi
SMOL_TILE_BOX_LPC8_MIN=$m python tools/bench_configs.py --configs BOX --iters 20 2>/dev/null | python -c "
import sys,json
print('lpc8_min=$m rows=$rows', ' '.join('%s=%.1f(%s)' % (d['config'], d['step_us'], d['kernel']) for d in map(json.loads, sys.stdin)))"
done; done
