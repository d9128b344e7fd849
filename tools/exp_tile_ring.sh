# k_tile's ring staging (round 6) against the two-buffer scheme, by ring depth and tile height (tuning build;
# profiles/r06_tile_ring.txt). Every row checked against the oracle (tests/bench_configs_checked.py).
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 15 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=${1:-R1.2,R1.5,R1.9,R2.5,R3,R3.5,R5,R7,BOX,BOX16,PHOTO,R1.5RGB,HDOWNVUP,U1.1,U1.5}
echo "two buffers (shipped)     $(f $C)"
for n in 2 4 8 12; do echo "ring N=$n                  $(SMOL_TILE_RING=$n f $C)"; done
echo "two buffers (shipped)     $(f $C)"
for r in 4 8 16; do echo "ring N=8 tile rows=$r      $(SMOL_TILE_RING=8 SMOL_TILE_ROWS=$r f $C)"; done
for r in 8 16; do echo "ring N=4 tile rows=$r      $(SMOL_TILE_RING=4 SMOL_TILE_ROWS=$r f $C)"; done
