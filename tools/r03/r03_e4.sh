run() { python tools/bench_configs.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)' % (d['config'], d['step_us'], d['frac_of_8TBps']) for d in map(json.loads, sys.stdin)))"; }
CF=BOX,R1.5,R3,R5,PHOTO
for lib in tuning rpf; do
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_$lib.so
for d in 0 1 5; do SMOL_TILE_DBG=$d run $CF $lib.dbg=$d; done
for d in 0 5; do SMOL_TILE_NO_SWZ=1 SMOL_TILE_DBG=$d run $CF $lib.noswz.dbg=$d; done
done
