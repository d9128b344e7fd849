f() { python tests/bench_configs_checked.py --configs $1 --iters 15 2>/dev/null | python -c "
import sys,json
print(' '.join('%.0f(%.3f)%s' % (d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
cfgs=""; cfgo=""
for m in 20 22 24 26 30 40; do
  w=$((1920*m/10/16*16)); h=$((1080*m/10)); cfgs="$cfgs,G:0:1920:1080:0:$w:$h:8"
  wo=$((1920*m/10/16*16+9)); cfgo="$cfgo,G:0:1920:1080:0:$wo:$h:8"
done
cfgs=${cfgs#,}; cfgo=${cfgo#,}
echo "x2.0 x2.2 x2.4 x2.6 x3.0 x4.0  (odd output widths: rows start on 4-byte boundaries)"
for lib in libsmolscale_amd.so libsmolscale_amd_upmis1.so libsmolscale_amd_upmis2.so libsmolscale_amd_upmis3.so libsmolscale_amd_upmis4.so libsmolscale_amd.so; do export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/$lib
echo "$lib odd     $(f $cfgo)"; done
