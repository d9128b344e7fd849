# the sweep's dip below ratio 1.5 (3840x2160 -> 2610 ... 2950 wide: 59-62 % where 2495 wide runs at 80 %): which rule makes it?
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 15 --placement ${2:-far} 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'].split(':')[5], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=G:2:3840:2160:2:2495:1403:12,G:2:3840:2160:2:2572:1446:12,G:2:3840:2160:2:2687:1511:12,G:2:3840:2160:2:2802:1576:12,G:2:3840:2160:2:2956:1662:12,G:2:3840:2160:2:3110:1749:12
echo "shipped far            $(f $C far)"
echo "shipped near           $(f $C near)"
echo "no ring                $(SMOL_NO_TILE_RING=1 f $C far)"
echo "ring N=8               $(SMOL_TILE_RING=8 f $C far)"
echo "ring N=2               $(SMOL_TILE_RING=2 f $C far)"
echo "CPL=1                  $(SMOL_TILE_CPL=1 f $C far)"
echo "rows=4                 $(SMOL_TILE_ROWS=4 f $C far)"
echo "rows=8                 $(SMOL_TILE_ROWS=8 f $C far)"
echo "rows=12                $(SMOL_TILE_ROWS=12 f $C far)"
echo "no XCD order           $(SMOL_TILE_NO_SWZ=1 f $C far)"
echo "write-through stores   $(SMOL_TILE_WT=1 f $C far)"
echo "shipped far            $(f $C far)"
