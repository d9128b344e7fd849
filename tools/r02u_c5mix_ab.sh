for rep in 1 2 3 4; do for lib in before now; do
  if [ $lib = now ]; then unset SMOLSCALE_AMD_LIB; else export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_before.so; fi
  python tests/bench_configs_checked.py --configs C5mix,C5 --iters 40 2>/dev/null | python -c "
import sys,json
print('$lib', ' '.join('%s=%.1f min %.1f (%d launches)%s' % (d['config'], d['step_us'], d['step_us_min'], d['launches_per_step'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"
done; done
