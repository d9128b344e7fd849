export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
mkdir -p gpurun_out/r02f
for rep in 1 2; do for by in 16 8 4 2 1; do
  SMOL_POW2_BY=$by python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('by=$by', d['roofline']['launch_us'], d['roofline']['frac'], d['roofline']['step_us_each'], d['verified_vs_oracle'])"
done; done | tee gpurun_out/r02f/pow2_block_rows.txt
