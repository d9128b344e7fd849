# A/B/C of library builds on one box, interleaved: bash tools/ab_libs.sh <configs> <lib.so> [<lib.so> ...]  (paths under smolscale_amd/)
cfgs=$1; shift
for rep in 1 2; do for lib in "$@"; do
  export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/$lib
  python tests/bench_configs_checked.py --configs $cfgs --iters 20 2>/dev/null | python -c "
import sys,json
print('$lib', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"
done; done
