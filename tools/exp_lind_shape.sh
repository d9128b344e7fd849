export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tools/bench_configs.py --configs $1 --iters 20 ${@:2} 2>/dev/null | python -c "
import sys,json
for d in map(json.loads, sys.stdin): print('  ', d['config'], d['kernel'], '%.1f us  %.3f' % (d['step_us'], d['frac_of_8TBps']))"; }
for n in 4 8 16; do echo "images $n"; f SRGBPHOTO,SRGBR7RGB --images $n; done
for r in 2 3 4 6 8; do echo "SMOL_DOWN_ROWS=$r (16 images)"; SMOL_DOWN_ROWS=$r f SRGBPHOTO,SRGBR7RGB --images 16; done
for w in 50 76 101 127; do echo "SMOL_LIND_W=$w (16 images)"; SMOL_LIND_W=$w f SRGBPHOTO --images 16; done
