for rep in 1 2; do for lib in tuning w3; do for d in 0 8; do
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_$lib.so
SMOL_TILE_DBG=$d python tools/bench_configs.py --configs R1.5,R3,R6,PHOTO,BOX --iters 20 2>/dev/null | python -c "
import sys,json
print('$lib dbg=$d', ' '.join('%s=%.1f' % (d['config'], d['step_us']) for d in map(json.loads, sys.stdin)))"
done; done; done
