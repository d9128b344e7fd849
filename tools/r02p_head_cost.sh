export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
for cfg in R1.5 R3; do for rows in 2 4 8 16; do
SMOL_TILE_ROWS=$rows SMOL_TILE_DBG=7 python tools/bench_configs.py --configs $cfg --iters 20 2>/dev/null | python -c "
import sys,json
print('rows=$rows dbg=7', ' '.join('%s=%.1f' % (d['config'], d['step_us']) for d in map(json.loads, sys.stdin)))"
done; done
