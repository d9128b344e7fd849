f() { python tests/bench_configs_checked.py --configs $1 --iters 15 --placement ${2:-far} 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'].split(':')[5], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
for n in 8 12 19 24; do
C=G:2:3840:2160:2:2495:1403:$n,G:2:3840:2160:2:2572:1446:$n,G:2:3840:2160:2:2687:1511:$n,G:2:3840:2160:2:2802:1576:$n,G:2:3840:2160:2:2956:1662:$n,G:2:3840:2160:2:3110:1749:$n
echo "$n frames far   $(f $C far)"
done
C=G:2:3840:2160:2:2495:1403:19,G:2:3840:2160:2:2572:1446:19,G:2:3840:2160:2:2687:1511:19,G:2:3840:2160:2:2802:1576:19,G:2:3840:2160:2:2956:1662:19,G:2:3840:2160:2:3110:1749:19
echo "19 frames off  $(f $C off)"
echo "19 frames near $(f $C near)"
