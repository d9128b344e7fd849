# box rotation A/B + other box ratios (tuning builds)
BOXES=BOX,R9,R10,R11
for rep in 1 2 3; do bash tools/ab_once.sh BOX libsmolscale_amd_norot.so libsmolscale_amd_tuning.so; done
