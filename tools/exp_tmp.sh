#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for rep in 1 2 3; do
 for ss in 1 0; do
  for th in 1 16; do
   echo "## C2 STREAM_STORES=$ss threads=$th"
   SMOL_HIP_STREAM_STORES=$ss tools/microbench/host_path $th 40 0
   echo "## 1080p STREAM_STORES=$ss threads=$th"
   SMOL_HIP_STREAM_STORES=$ss tools/microbench/host_path $th 100 0 1920 1080 1280 720
   echo "## up STREAM_STORES=$ss threads=$th"
   SMOL_HIP_STREAM_STORES=$ss tools/microbench/host_path $th 100 0 1024 1024 2048 2048
  done
 done
done
} > gpurun_out/mapped_stream_stores.txt 2>&1
