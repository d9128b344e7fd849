# k_tile's tile height for magnifications of OTHER pixel types (24-bit rows, unassociated alpha on either side), outputs in rotation
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 12 --placement far 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (':'.join(d['config'].split(':')[1:7:2]) if d['config'].startswith('G:') else d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=G:8:3840:2160:8:4185:2354:8,G:8:3840:2160:8:4953:2786:8,G:8:3840:2160:0:4953:2786:8,G:4:3840:2160:0:4185:2354:8,G:4:3840:2160:0:4953:2786:8,G:0:3840:2160:5:4953:2786:8,G:0:3840:2160:8:4953:2786:8,G:8:3840:2160:8:6335:3563:6,G:4:3840:2160:0:6335:3563:6
echo "shipped (12)  $(f $C)"
for r in 4 5 6 8; do echo "rows=$r        $(SMOL_TILE_ROWS=$r f $C)"; done
echo "shipped (12)  $(f $C)"
