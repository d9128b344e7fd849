# ratio 1.5 (0H bilinear both axes, 4 columns per lane): tile rows x store policy x ablation, tuning builds
for lib in tuning nt plain; do for rows in 2 4 8 16 32; do for d in 0 2; do
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_$lib.so
SMOL_TILE_ROWS=$rows SMOL_TILE_DBG=$d python tools/bench_configs.py --configs R1.5 --iters 20 2>/dev/null | python -c "
import sys,json
print('$lib rows=$rows dbg=$d', ' '.join('%s=%.1f' % (d['config'], d['step_us']) for d in map(json.loads, sys.stdin)))"
done; done; done
