export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tools/bench_configs.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)' % (d['config'], d['step_us'], d['frac_of_8TBps']) for d in map(json.loads, sys.stdin)))"; }
CF=BOX,R1.2,R1.5,R3,R5,PHOTO,PNG,HDOWNVUP
for d in 0 1 2 3 4 5 7; do SMOL_TILE_DBG=$d run $CF dbg=$d; done
