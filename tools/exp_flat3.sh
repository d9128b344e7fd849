# k_lind's flat three-channel flavour against k_tile on 24-bit rows of the 64bpp class (tuning build; profiles/r05_flat3_ab.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'][:22], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=PHOTO,PHOTOODD,G:8:4000:3000:0:1600:1200:12,G:8:4000:3000:8:1900:1425:12,G:8:4000:3000:8:1000:750:12,G:8:4000:3000:9:700:525:12,G:8:4000:3000:4:1333:1000:12
D=R1.5RGB,U4RGB,G:8:4000:3000:8:3000:2250:8,G:8:4000:3000:8:2100:1575:12,G:8:1920:1080:8:3840:2160:6,G:8:2000:1500:0:1200:2250:8
for rep in 1 2; do
echo "k_tile                 $(SMOL_NO_FLAT3=1 f $C)"
echo "flat3 (halvings >= 1)  $(f $C)"
done
echo "--- no halving, magnification, mixed axes"
for rep in 1 2; do
echo "k_tile / k_up          $(SMOL_NO_FLAT3=1 f $D)"
echo "flat3 everywhere       $(SMOL_FLAT3_ALL=1 f $D)"
done
