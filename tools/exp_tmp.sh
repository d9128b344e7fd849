show() { python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ', d['rows_per_call'], ' '.join('%s: %.2fx (%.1f us)' % (k, d[k]['x_whole'], d[k]['us_per_call']) for k in ('1','8','64','all')))"; }
echo default; python tools/bench_rows_per_call.py 2>/dev/null | show
echo SMOL_HIP_RESIDENT_SRC=1; SMOL_HIP_RESIDENT_SRC=1 python tools/bench_rows_per_call.py 2>/dev/null | show
echo SMOL_HIP_ZEROCOPY_KB=1024; SMOL_HIP_ZEROCOPY_KB=1024 python tools/bench_rows_per_call.py 2>/dev/null | show
echo SMOL_HIP_ZEROCOPY_KB=4096; SMOL_HIP_ZEROCOPY_KB=4096 python tools/bench_rows_per_call.py 2>/dev/null | show
echo SMOL_HIP_READAHEAD=0; SMOL_HIP_READAHEAD=0 python tools/bench_rows_per_call.py 2>/dev/null | show
