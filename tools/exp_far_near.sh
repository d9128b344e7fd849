# round 6: the same configs with their destinations 32 GB behind their sources inside one region ("far": another zone of the device's
# memory), right behind them ("near": the same zone), and in separate allocations wherever the allocator puts them ("off");
# profiles/r06_headline_modes.txt, last section. Every row bit-exact against the oracle.
f() { python tests/bench_configs_checked.py --configs $1 --iters 20 --placement $2 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=${1:-C1,C2,C3,C4,C5,C5mix,BOX,C2CROP4,R1.5,R2.5,R5,U1.1,U3,PHOTO,PNG,THUMB,P16,SRGB,SRGBHALF}
for p in far near off far near; do echo "$p   $(f $C $p)"; done
