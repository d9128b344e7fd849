for pair in "0 0" "4 0" "0 4" "0 8" "4 8" "8 0" "8 4" "8 8" "4 4"; do
  set -- $pair
  python tests/bench_configs_checked.py --configs C5 --tin $1 --tout $2 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$1->$2', d['kernel'], d['step_us'], d['GBps'], d['bit_exact_first_rows_vs_oracle'])"
done
