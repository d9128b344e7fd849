export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
for cfg in ${CFGS:-R3 PHOTO BOX}; do for d in 0 1 2 4 3 5 7; do
SMOL_TILE_DBG=$d python tools/bench_configs.py --configs $cfg --iters 20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$cfg dbg=$d', d['kernel'], d['step_us'], d['frac_of_8TBps'])
"
done; done
