export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
CF=BOX,R1.2,R1.5,R1.7,R1.9,R2.5,R3,R3.5,R5,R6,R7,PHOTO,PNG,R1.5RGB,R1.5U,HDOWNVUP
SMOL_TILE_NO_SWZ=1 run $CF noswz
run $CF swz
SMOL_TILE_WG1=1 run $CF swz+wg1
