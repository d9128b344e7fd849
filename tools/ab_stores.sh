for lib in prod nt prod nt; do
  if [ $lib = nt ]; then export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_nt.so; else unset SMOLSCALE_AMD_LIB; fi
  python tools/bench_configs.py --configs C1,C2,C3,C4,C5,C5mix,BOX,R3,P16,THUMB --iters 20 2>/dev/null | python -c "
import sys,json
print('$lib', ' '.join('%s=%.1f(%.3f)' % (d['config'], d['step_us'], d['frac_of_8TBps']) for d in map(json.loads, sys.stdin)))"
done
