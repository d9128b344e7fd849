# k_tile's ring loop: which issue port does a class sit on? Sixteen more scalar / vector instructions per source row (tuning build:
# SMOL_TILE_DBG 8 / 16 / 24), no setprio, against the shipped loop (profiles/r06_sq_tile_classes.txt).
cd "$GRAFT_REPO_ROOT" || exit 1
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tools/bench_configs.py --configs $1 --iters 20 --placement far 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f' % (d['config'], d['step_us']) for d in map(json.loads, sys.stdin)))"; }
C=${1:-R1.5,R2.5,R3,R5,R7,PHOTO,PNG,U1.1,U1.5}
echo "as shipped                    $(f $C)"
echo "+16 scalar per row            $(SMOL_TILE_DBG=8 f $C)"
echo "+16 vector per row            $(SMOL_TILE_DBG=16 f $C)"
echo "+16 of each                   $(SMOL_TILE_DBG=24 f $C)"
echo "as shipped                    $(f $C)"
