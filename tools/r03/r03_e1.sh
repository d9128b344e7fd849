mkdir -p gpurun_out/r03a
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
CF=BOX,R1.2,R1.5,R3,R5,PHOTO,PNG
run $CF base
for b in 2048 2560 3072 6144; do SMOL_TILE_LDS_BUDGET=$b run $CF budget=$b; done
for r in 2 3 6 8 12; do SMOL_TILE_ROWS=$r run $CF rows=$r; done
SMOL_TILE_NO_SINGLE=1 run $CF nosingle
SMOL_TILE_WG1=1 run $CF wg1
