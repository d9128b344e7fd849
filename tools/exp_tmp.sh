#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for ck in 1024 2048 4096 8192; do
  echo "## mapped CHUNK_KB=$ck"
  SMOL_HIP_TRACE=1 SMOL_HIP_MAPPED_CHUNK_KB=$ck tools/microbench/host_path 1 6 0 2>&1 | tail -4
done
echo "## DMA road"
SMOL_HIP_TRACE=1 SMOL_HIP_MAPPED=0 tools/microbench/host_path 1 6 0 2>&1 | tail -4
for ct in 2 4 16; do
  echo "## mapped COPY_THREADS=$ct"
  SMOL_HIP_TRACE=1 SMOL_HIP_COPY_THREADS=$ct tools/microbench/host_path 1 6 0 2>&1 | tail -3
done
} > gpurun_out/mapped_trace.txt 2>&1
for f in default MAPPED_CHUNK_KB=64 BOUNCE_KB=64 MAPPED=0; do
  case $f in default) e="";; *) e="SMOL_HIP_$f";; esac
  env $e python3 tools/fuzz/fuzz_host_road.py 7 400 2>&1 | tail -3
done > gpurun_out/fuzz_host_road.txt 2>&1
cat gpurun_out/fuzz_host_road.txt
