f() { python tests/bench_configs_checked.py --configs $1 --iters 15 --placement $2 $3 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)x%s%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], d.get('destination_sets'), '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=G:0:4096:4096:0:4096:4096:6,G:0:4096:4096:0:4096:2048:6,G:0:4096:4096:0:2048:4096:6,G:0:4096:4096:0:4095:4095:6
echo "far rot     $(f $C far)"
echo "near rot    $(f $C near)"
echo "far one set $(f $C far --no-rotate)"
echo "off rot     $(f $C off)"
