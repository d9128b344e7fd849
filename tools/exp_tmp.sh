cd "$GRAFT_REPO_ROOT" || exit 1
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'][:26], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
A=U2UNASSOC,G:0:1920:1080:4:4320:2430:4,G:0:1920:1080:4:4993:2809:4,G:0:1920:1080:5:5760:3240:4,G:0:1920:1080:5:5763:3241:4,G:0:1920:1080:4:6336:3564:4
for rep in 1 2; do
echo "k_tile             $(SMOL_NO_UP_U4=1 f $A)"
echo "k_up (20 rows)     $(SMOL_UP_ALWAYS=1 f $A)"
done
