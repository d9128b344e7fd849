# k_tile's tile height for magnification x1.4 ... x2.2 and minification 1.2 ... 2 (four columns per lane), outputs in rotation, arenas placed
# (tuning build: SMOL_TILE_ROWS; profiles/r06_tile_rows_near1.txt)
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 12 --placement far 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'].split(':')[5] if d['config'].startswith('G:') else d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=G:2:3840:2160:2:5414:3045:6,U1.5,G:2:3840:2160:2:6335:3563:6,G:2:3840:2160:2:6950:3909:4,U2,G:2:3840:2160:2:8255:4643:4,G:2:3840:2160:2:2956:1662:8,R1.35,R1.5,G:2:3840:2160:2:2534:1425:8,R1.7,G:2:3840:2160:2:2188:1230:8,R1.9
echo "shipped   $(f $C)"
for r in 4 5 6 8 10 12; do echo "rows=$r    $(SMOL_TILE_ROWS=$r f $C)"; done
echo "shipped   $(f $C)"
