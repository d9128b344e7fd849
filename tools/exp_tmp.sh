f() { python tools/bench_configs.py --configs $1 --iters 20 ${@:2} 2>/dev/null | python -c "
import sys,json
for d in map(json.loads, sys.stdin): print('  ', d['config'], d['kernel'], '%.1f us  %.3f' % (d['step_us'], d['frac_of_8TBps']))"; }
for lib in libsmolscale_amd_before.so libsmolscale_amd.so; do export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/$lib; echo $lib
echo " noise 16"; f SRGBPHOTO,SRGBR7RGB --images 16
echo " natural 16"; f SRGBPHOTO,SRGBR7RGB --natural --images 16
done
