# smol_hip_memory_zones() in N fresh processes (240 GB region, windows of 2 GB and of 512 MB): how many calls find three kinds; the sweeps'
# times (SMOL_HIP_ZONES_TRACE) of the processes that did not are kept in gpurun_out/zones_fail_<i>.txt
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
N=${1:-16}
for i in $(seq 1 $N); do
  SMOL_HIP_ZONES_TRACE=1 ZONE_MAP_LIBRARY_ONLY=1 python tools/exp_zone_map.py 240 2048 > gpurun_out/zr_out.txt 2> gpurun_out/zr_err.txt
  grep -v amdgpu gpurun_out/zr_out.txt | grep -v "^region"
  if [ "$(grep -c '3 kinds' gpurun_out/zr_out.txt)" != "2" ]; then cp gpurun_out/zr_err.txt gpurun_out/zones_fail_$i.txt; fi
done
