export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
for cpl in 1 4; do for d in 0; do
SMOL_TILE_CPL=$cpl SMOL_TILE_DBG=$d python tools/bench_configs.py --configs R1.2,R1.5,R1.9,R1.5RGB,R1.5U,HDOWNVUP --iters 20 2>/dev/null | python -c "
import sys,json
print('cpl=$cpl dbg=$d', ' '.join('%s=%.1f' % (d['config'], d['step_us']) for d in map(json.loads, sys.stdin)))"
done; done
