# A/B two builds of the library on the same box: SMOLSCALE_AMD_LIB picks the .so
CFG=${1:-C3,C4,BOX,R3,PHOTO,PNG}
for rep in 1 2; do
for lib in gpurun_old.so smolscale_amd/libsmolscale_amd.so; do
  SMOLSCALE_AMD_LIB=$PWD/$lib python tools/bench_configs.py --configs $CFG --iters 40 | python -c "
import sys,json
print('$lib', ' '.join('%s %.1f' % (json.loads(l)['config'], json.loads(l)['step_us']) for l in sys.stdin))"
done; done
