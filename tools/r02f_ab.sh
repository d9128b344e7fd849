for rep in 1 2 3; do
  for lib in prod tuning; do
    if [ $lib = tuning ]; then export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so; else unset SMOLSCALE_AMD_LIB; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib', d['roofline']['launch_us'], d['roofline']['frac'], d['roofline']['step_us_each'])"
  done
done
