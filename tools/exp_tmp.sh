#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
echo "## CPU backend, SIMD off, 1 thread (the generic 32-bit-lane road: every class before round 5 except the first two rows)"
SMOL_CPU_SIMD=0 SMOL_CPU_THREADS=1 tools/microbench/cpu_backend
echo "## CPU backend, 1 thread"
SMOL_CPU_THREADS=1 tools/microbench/cpu_backend
echo "## CPU backend, 8 threads"
SMOL_CPU_THREADS=8 tools/microbench/cpu_backend
echo "## CPU backend, default threads (32)"
tools/microbench/cpu_backend
nproc; grep -m1 "model name" /proc/cpuinfo
} > gpurun_out/cpu_backend_classes.txt 2>&1
