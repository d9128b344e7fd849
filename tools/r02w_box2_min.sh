export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
for m in 12 8; do for g in none SMOL_BOXH_XGROUP SMOL_BOXH_YGROUP; do
env SMOL_BOX2_MIN=$m $( [ $g != none ] && echo $g=1 ) python tools/bench_configs.py --configs BOX,THUMB --iters 20 2>/dev/null | python -c "
import sys,json
print('box2_min=$m $g', ' '.join('%s=%.1f(%s)' % (d['config'], d['step_us'], d['kernel']) for d in map(json.loads, sys.stdin)))"
done; done
