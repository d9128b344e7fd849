f() { python tools/bench_configs.py --configs $1 --iters 20 ${@:2} 2>/dev/null | python -c "
import sys,json
for d in map(json.loads, sys.stdin): print('  ', d['config'], d['kernel'], '%.1f us  %.3f' % (d['step_us'], d['frac_of_8TBps']))"; }
echo "white noise"; f SRGBPHOTO,SRGBR7RGB,SRGBUP24,SRGBHALF,SRGBTHUMB,SRGBBOX,SRGBUP
echo "--natural"; f SRGBPHOTO,SRGBR7RGB,SRGBUP24,SRGBHALF,SRGBTHUMB,SRGBBOX,SRGBUP --natural
echo "--natural --images 16"; f SRGBPHOTO,SRGBR7RGB --natural --images 16
