# k_tile behind one horizontal halving: where the time goes, by ablation (tuning build; results are NOT bit-exact with parts switched off).
# SMOL_TILE_DBG: 1 no taps (rows staged and stored, nothing computed), 2 no staging (no loads), 4 no stores; sums of them.
cd "$GRAFT_REPO_ROOT" || exit 1
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tools/bench_configs.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f' % (d['config'], d['step_us']) for d in map(json.loads, sys.stdin)))"; }
C=R1.5,R2.5,R3,R5,PHOTO,PNG,BOX
echo "as shipped                    $(f $C)"
echo "no taps                       $(SMOL_TILE_DBG=1 f $C)"
echo "no loads                      $(SMOL_TILE_DBG=2 f $C)"
echo "no stores                     $(SMOL_TILE_DBG=4 f $C)"
echo "no taps, no stores            $(SMOL_TILE_DBG=5 f $C)"
echo "no loads, no stores           $(SMOL_TILE_DBG=6 f $C)"
echo "loop skeleton only            $(SMOL_TILE_DBG=7 f $C)"
echo "as shipped                    $(f $C)"
