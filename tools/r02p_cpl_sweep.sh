# k_tile: columns per lane (SMOL_TILE_CPL) x LDS staging budget, tuning build only
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
for cfg in R1.5 R3 R6 PHOTO PNG; do for cpl in 1 4; do for b in 4096 8192 16384; do
SMOL_TILE_CPL=$cpl SMOL_TILE_LDS_BUDGET=$b python tools/bench_configs.py --configs $cfg --iters 20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$cfg cpl=$cpl budget=$b', d['kernel'], d['step_us'], d['frac_of_8TBps'])
"
done; done; done
