# k_tile levers for ratios 2.3-4 (one halving) and the box: columns per lane, LDS budget, tile rows (tuning build; profiles/r05_tile_levers.txt)
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tools/bench_configs.py --configs $1 --iters 15 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)' % (d['config'], d['step_us'], d['frac_of_8TBps']) for d in map(json.loads, sys.stdin)))"; }
C=R2.5,R3,R3.5,R5,PHOTO
echo "shipped                         $(f $C)"
for b in 8192 16384; do echo "CPL=4 budget=$b                $(SMOL_TILE_CPL=4 SMOL_TILE_LDS_BUDGET=$b f $C)"; done
for b in 2048 3072 6144 8192; do echo "CPL=1 budget=$b                $(SMOL_TILE_LDS_BUDGET=$b f $C)"; done
for r in 2 3 6 8; do echo "tile rows=$r                     $(SMOL_TILE_ROWS=$r f $C)"; done
echo "no XCD order                    $(SMOL_TILE_NO_SWZ=1 f $C)"
echo "single staging buffer off       $(SMOL_TILE_NO_SINGLE=1 f $C)"
echo "shipped                         $(f $C)"
echo "BOX shipped                     $(f BOX,BOX16,THUMB)"
for b in 2048 3072 6144 8192; do echo "BOX budget=$b                   $(SMOL_TILE_LDS_BUDGET=$b f BOX,BOX16)"; done
for r in 3 5 10 14; do echo "BOX tile rows=$r                 $(SMOL_TILE_ROWS=$r f BOX,BOX16)"; done
