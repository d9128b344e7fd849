for rep in 1 2; do
for args in "--tout 0" "--tout 1" "--tout 4" "--tout 8" "--tin 4 --tout 0" "--tin 8 --tout 0" "--tin 4 --tout 4"; do
python tools/bench_configs.py --configs C2,C5 $args --iters 20 2>/dev/null | python -c "
import sys,json
print('$args', ' '.join('%s=%.1f(%.3f,%s)' % (d['config'], d['step_us'], d['frac_of_8TBps'], d['kernel']) for d in map(json.loads, sys.stdin)))"
done; done
