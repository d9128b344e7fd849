export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 20 ${@:2} 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=SRGBHALF,SRGBVCOPY,SRGBHDVU,SRGBU1.1,SRGBR6
for i in 1 2; do
echo "chain      $(f $C)"
echo "fused tab  $(SMOL_LIND_TRI=1 f $C)"
done
echo "16 frames: chain $(f SRGBHALF --images 16)  fused $(SMOL_LIND_TRI=1 f SRGBHALF --images 16)"
echo "natural:   chain $(f SRGBHALF --natural)  fused $(SMOL_LIND_TRI=1 f SRGBHALF --natural)"
echo "opaque:    chain $(f SRGBHALF --opaque)  fused $(SMOL_LIND_TRI=1 f SRGBHALF --opaque)"
