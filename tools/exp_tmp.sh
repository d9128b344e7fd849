cd "$GRAFT_REPO_ROOT" || exit 1
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'][:22], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=PHOTO,PHOTOODD,G:8:4000:3000:0:1600:1200:12,G:8:4000:3000:8:1900:1425:12,G:8:4000:3000:9:700:525:12,G:8:4000:3000:4:1333:1000:12
echo "k_tile          $(SMOL_NO_FLAT3=1 f $C)"
for r in 1 2 3 4 5 6; do echo "flat3 rows=$r    $(SMOL_DOWN_ROWS=$r f $C)"; done
echo "k_tile          $(SMOL_NO_FLAT3=1 f $C)"
echo "gamma rows (k_lind as shipped: 4 rows behind a vertical halving, else 8)"
G=SRGBPHOTO,SRGBR7RGB,SRGBHALF,SRGBPNG,SRGBR6,P16HALF
echo "shipped         $(f $G)"
for r in 2 3 4 6 8; do echo "rows=$r          $(SMOL_DOWN_ROWS=$r f $G)"; done
