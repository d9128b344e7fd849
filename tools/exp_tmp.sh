cd "$GRAFT_REPO_ROOT" || exit 1
{
echo "# randomized GPU parity runs on the round-5 final binary (commit d6efaba kernels: k_lind two-row tiles behind two halvings, k_up with 24-bit output; the mapped host road), against the oracle"
python3 tools/fuzz/gpu_fuzz.py 501 100000 2>&1 | tail -2
python3 tools/fuzz/gpu_fuzz_r05.py 502 60000 2>&1 | tail -2
python3 tools/fuzz/fuzz_lind3.py 503 2>&1 | tail -1
for e in "" "SMOL_HIP_MAPPED_CHUNK_KB=64" "SMOL_HIP_BOUNCE_KB=64" "SMOL_HIP_MAPPED=0"; do env $e python3 tools/fuzz/fuzz_host_road.py 504 1500 2>&1 | tail -1; done
} > gpurun_out/r05final_gpu_fuzz.txt 2>&1
cat gpurun_out/r05final_gpu_fuzz.txt
