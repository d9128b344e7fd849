# k_tile against k_up (and k_up with taller tiles) for magnifications x1.8 .. x3.2 of 1920x1080 RGBA8 premultiplied,
# aligned and odd output widths (profiles/r05_up_handover.txt); tuning build
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tools/bench_configs.py --configs $1 --iters 15 2>/dev/null | python -c "
import sys,json
print(' '.join('%.0f(%.3f)' % (d['step_us'], d['frac_of_8TBps']) for d in map(json.loads, sys.stdin)))"; }
cfgs=""; cfgo=""
for m in 18 19 20 21 22 23 24 25 26 28 30 32; do
  w=$((1920*m/10/16*16)); h=$((1080*m/10)); cfgs="$cfgs,G:0:1920:1080:0:$w:$h:8"
  wo=$((1920*m/10/16*16+9)); cfgo="$cfgo,G:0:1920:1080:0:$wo:$h:8"
done
cfgs=${cfgs#,}; cfgo=${cfgo#,}
echo "magnification          x1.8 x1.9 x2.0 x2.1 x2.2 x2.3 x2.4 x2.5 x2.6 x2.8 x3.0 x3.2   step us (fraction of 8 TB/s), 8 frames of 1920x1080"
echo "aligned  shipped rule  $(f $cfgs)"
echo "aligned  k_tile        $(SMOL_NO_UP=1 f $cfgs)"
echo "aligned  k_up 8 rows   $(SMOL_UP_ALWAYS=1 f $cfgs)"
echo "aligned  k_up 6 rows   $(SMOL_UP_ALWAYS=1 SMOL_UP_ROWS=6 f $cfgs)"
echo "aligned  k_up 12 rows  $(SMOL_UP_ALWAYS=1 SMOL_UP_ROWS=12 f $cfgs)"
echo "aligned  k_up 16 rows  $(SMOL_UP_ALWAYS=1 SMOL_UP_ROWS=16 f $cfgs)"
echo "odd      shipped rule  $(f $cfgo)"
echo "odd      k_tile        $(SMOL_NO_UP=1 f $cfgo)"
echo "odd      k_up 8 rows   $(SMOL_UP_ALWAYS=1 f $cfgo)"
echo "odd      k_up 6 rows   $(SMOL_UP_ALWAYS=1 SMOL_UP_ROWS=6 f $cfgo)"
echo "odd      k_up 12 rows  $(SMOL_UP_ALWAYS=1 SMOL_UP_ROWS=12 f $cfgo)"
