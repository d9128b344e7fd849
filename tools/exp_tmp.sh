cd "$GRAFT_REPO_ROOT" || exit 1
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'][:26], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
A=G:0:1920:1080:4:7680:4320:4,G:1:1920:1080:6:7683:4321:4,G:8:1920:1080:4:7680:4320:4,G:0:960:540:4:7680:4320:4,G:0:1920:1080:5:6720:3780:4
for rep in 1 2; do
echo "shipped (8 rows)  $(f $A)"
for r in 12 16 20 24 32; do echo "rows=$r   $(SMOL_UP_ROWS=$r f $A)"; done
done
