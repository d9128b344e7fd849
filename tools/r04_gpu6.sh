cd $GRAFT_REPO_ROOT
for c in SRGBHALF SRGBPHOTO; do bash tools/sq_counters.sh r04e $c > /dev/null 2>&1; cat gpurun_out/r04e/sq_$c.txt; done
