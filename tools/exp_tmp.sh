cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "ramp on : $(tools/microbench/host_path 1 30 0 | tail -1)   16 callers: $(tools/microbench/host_path 16 8 0 | tail -1)"
echo "ramp off: $(SMOL_HIP_RAMP=0 tools/microbench/host_path 1 30 0 | tail -1)   16 callers: $(SMOL_HIP_RAMP=0 tools/microbench/host_path 16 8 0 | tail -1)"
done
echo "pinned  : $(tools/microbench/host_path 1 30 1 | tail -1)"
SMOL_HIP_TRACE=1 tools/microbench/host_path 1 6 0 2>&1 | grep "strip rows" | tail -2
