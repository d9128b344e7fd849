cd "$GRAFT_REPO_ROOT" || exit 1
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'][:26], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
G=SRGBR6,SRGBR7RGB,G:4:4096:4096:4:683:683:4,G:4:4096:4096:5:585:585:4:srgb,G:8:4000:3000:8:571:428:8:srgb,G:0:1000:4000:0:2000:600:4:srgb,G:4:4096:4096:0:900:900:4:srgb,G:9:4000:3000:0:800:600:8:srgb
for rep in 1 2; do
echo "rows=4 (round 4)  $(SMOL_DOWN_ROWS=4 f $G)"
echo "rows=2 (now)      $(f $G)"
echo "rows=3            $(SMOL_DOWN_ROWS=3 f $G)"
echo "rows=1            $(SMOL_DOWN_ROWS=1 f $G)"
done
