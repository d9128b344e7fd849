cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/fuzz/gpu_fuzz.py 81 12000 > gpurun_out/r04j/fuzz.txt 2>&1; tail -3 gpurun_out/r04j/fuzz.txt | cut -c1-800
python tests/bench_configs_checked.py --configs SRGBHDVU,SRGBVCOPY,SRGBBOXBIL,P16BOXBIL,P16HDVU,SRGBHALF,SRGBPHOTO --iters 10 2>/dev/null | python -c "
import sys,json
for d in map(json.loads, sys.stdin): print(d['config'], d['kernel'], d['step_us'], d['frac_of_8TBps'], d.get('bit_exact_first_rows_vs_oracle'))"
