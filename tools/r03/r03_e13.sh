export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$2', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
CF=C3,R1.03,R1.1,R1.2,R1.35,R1.5,R1.7,R1.9,R1.5RGB,R1.5U,HDOWNVUP,PHOTO,U1.1,U1.37,U1.5,U1.75,U2,U2.6,U3,P16UP,SRGBUP,P16HALF
for rep in 1 2; do
SMOL_VEC16_ALIGNED=1 SMOL_TILE_CPL1_BELOW_175=1 SMOL_UP_ALWAYS=1 run $CF old
run $CF new
done
