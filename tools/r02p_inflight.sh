export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
for b in 1700 2500 3300 4096 5000; do
SMOL_TILE_LDS_BUDGET=$b python tools/bench_configs.py --configs R3,R6,R1.5 --iters 20 2>/dev/null | python -c "
import sys,json
print('budget=$b', ' '.join('%s=%.1f' % (d['config'], d['step_us']) for d in map(json.loads, sys.stdin)))"
done
