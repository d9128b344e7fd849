# A/B of two builds of the library on one box: bash tools/ab_generic.sh <configs> (libs: smolscale_amd/libsmolscale_amd_before.so vs the current one)
for rep in 1 2; do for lib in before now; do
  if [ $lib = now ]; then unset SMOLSCALE_AMD_LIB; else export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_before.so; fi
  python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print('$lib', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"
done; done
