cd $GRAFT_REPO_ROOT
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -5
(for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Power\|Average" | tr '\n' ' '; echo; sleep 0.15; done) > gpurun_out/r04f/clocks.txt &
python tools/bench_configs.py --configs C5mix --iters 3000 2>/dev/null | python -c "
import sys,json
for d in map(json.loads, sys.stdin): print(d['config'], d['step_us'], d['step_us_min'], d['frac_of_8TBps'])"
wait
sort gpurun_out/r04f/clocks.txt | uniq -c | sort -rn | head -12
