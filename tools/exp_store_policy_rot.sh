# k_tile's store policy once the outputs really go to memory (outputs in rotation, arenas placed): write-back (shipped for rows that
# start anywhere), write-through (SMOL_TILE_WT=1), and the ring / tile-height knobs again (tuning build; profiles/r06_store_policy_rot.txt)
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 15 --placement far 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=${1:-R1.03,R1.2,R1.5,R1.9,R2.5,R3,U1.1,U1.37,U1.5,U2,S4K91ODD,S4K109ODD,HDOWNVUP,PHOTO}
echo "shipped                $(f $C)"
echo "write-through          $(SMOL_TILE_WT=1 f $C)"
echo "no write-through       $(SMOL_TILE_NO_WT=1 f $C)"
echo "no ring                $(SMOL_NO_TILE_RING=1 f $C)"
echo "ring N=8               $(SMOL_TILE_RING=8 f $C)"
echo "rows=4                 $(SMOL_TILE_ROWS=4 f $C)"
echo "rows=12                $(SMOL_TILE_ROWS=12 f $C)"
echo "shipped                $(f $C)"
