# k_tile's tile height for scalings near 1 (ratio 0.75 ... 1.4: a third of the reference's sweep), outputs in rotation, arenas placed
# (tuning build: SMOL_TILE_ROWS; profiles/r06_tile_rows_near1.txt)
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 12 --placement far 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'].split(':')[5] if d['config'].startswith('G:') else d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=G:2:3840:2160:2:2802:1576:8,G:2:3840:2160:2:3110:1749:8,G:2:3840:2160:2:3494:1965:8,G:2:3840:2160:2:3724:2094:8,G:2:3840:2160:2:3955:2224:8,G:2:3840:2160:2:4185:2354:8,G:2:3840:2160:2:4492:2526:8,G:2:3840:2160:2:4953:2786:8,G:2:3840:2160:2:5414:3045:6,R1.2,R1.03,U1.1
echo "shipped   $(f $C)"
for r in 3 4 5 6 8; do echo "rows=$r    $(SMOL_TILE_ROWS=$r f $C)"; done
echo "shipped   $(f $C)"
