# k_tile at ratios 2.5-7 (one / two halvings, 32-bit premultiplied): columns per lane x tile height, WITH ring staging and the outputs in
# rotation -- round 5 measured two and four columns per lane against one under the two-buffer scheme (profiles/r05_tile_cpl2.txt,
# r05_tile_levers.txt). Tuning build: SMOL_TILE_CPL, SMOL_TILE_ROWS.
cd "$GRAFT_REPO_ROOT" || exit 1
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 20 --placement far 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f%s' % (d['config'], d['step_us'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=${1:-R2.5,R3,R3.5,R5,R6,R7}
echo "as shipped                    $(f $C)"
for cpl in 1 2 4; do
  for rows in 0 2 3 4 6 8; do
    if [ $rows = 0 ]; then echo "cpl $cpl rows default          $(SMOL_TILE_CPL=$cpl f $C)"; else echo "cpl $cpl rows $rows                $(SMOL_TILE_CPL=$cpl SMOL_TILE_ROWS=$rows f $C)"; fi
  done
done
echo "as shipped                    $(f $C)"
