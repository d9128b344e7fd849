cd "$GRAFT_REPO_ROOT" || exit 1
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f' % (d['config'][2:22], d['step_us']) for d in map(json.loads, sys.stdin)))"; }
# the one-column-into-24-bit-rows classes: unassociated 2.5 / 3.8, premultiplied 3 / 3.8, RGB8 3.8; plus PHOTO (4 columns) and PNG
U=G:4:4000:3000:8:1600:1200:10,G:4:4000:3000:8:1053:790:10
P=G:0:4000:3000:8:1333:1000:10,G:0:4000:3000:8:1053:790:10
R=G:8:4000:3000:8:1053:790:12,G:8:4000:3000:8:1600:1200:12
for rows in default 2 3 4 5 6 8; do
  if [ $rows = default ]; then unset SMOL_TILE_ROWS; else export SMOL_TILE_ROWS=$rows; fi
  echo "rows=$rows  $(f $U,$P,$R,PNG)"
done
