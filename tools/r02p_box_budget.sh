export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
for b in 4096 6000 8192 12288 16384 24576; do
SMOL_TILE_LDS_BUDGET=$b python tools/bench_configs.py --configs BOX,THUMB,R6 --iters 20 2>/dev/null | python -c "
import sys,json
print('budget=$b', ' '.join('%s=%.1f' % (d['config'], d['step_us']) for d in map(json.loads, sys.stdin)))"
done
