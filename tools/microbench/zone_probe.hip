// zone_probe: what a pure streaming READ and a pure streaming WRITE of a window of an existing device allocation take, as a shared
// library for tools/exp_zone_map.py: are the slow and fast zones of device memory (profiles/r06_zone_map.txt) a property of the
// memory itself, and of which direction?
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/microbench/libzone_probe.so tools/microbench/zone_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));

// 4 independent 16-byte non-temporal loads per lane, a workgroup of 256 lanes covers 16 KiB
__global__ void __launch_bounds__ (256) k_read (const u32x4 *p, uint32_t *sink)
{
    const u32x4 *q = p + (size_t) blockIdx.x * 1024 + threadIdx.x;
    u32x4 a = __builtin_nontemporal_load (q), b = __builtin_nontemporal_load (q + 256);
    u32x4 c = __builtin_nontemporal_load (q + 512), d = __builtin_nontemporal_load (q + 768);
    a ^= b; c ^= d; a ^= c;
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345679u) *sink = 1;
}
__global__ void __launch_bounds__ (256) k_write (u32x4 *p, uint32_t v)
{
    u32x4 *q = p + (size_t) blockIdx.x * 1024 + threadIdx.x;
    const u32x4 x = { v, v + 1, blockIdx.x, threadIdx.x };
    __builtin_nontemporal_store (x, q); __builtin_nontemporal_store (x, q + 256);
    __builtin_nontemporal_store (x, q + 512); __builtin_nontemporal_store (x, q + 768);
}

// 16 KiB per workgroup from s to d, non-temporal both ways
__global__ void __launch_bounds__ (256) k_copy (const u32x4 *s, u32x4 *d)
{
    const u32x4 *q = s + (size_t) blockIdx.x * 1024 + threadIdx.x;
    u32x4 *o = d + (size_t) blockIdx.x * 1024 + threadIdx.x;
    u32x4 a = __builtin_nontemporal_load (q), b = __builtin_nontemporal_load (q + 256);
    u32x4 c = __builtin_nontemporal_load (q + 512), e = __builtin_nontemporal_load (q + 768);
    __builtin_nontemporal_store (a, o); __builtin_nontemporal_store (b, o + 256);
    __builtin_nontemporal_store (c, o + 512); __builtin_nontemporal_store (e, o + 768);
}
// 64 KiB per workgroup from s, 16 KiB to d (the headline's shape: four bytes read per byte written)
__global__ void __launch_bounds__ (256) k_reduce4 (const u32x4 *s, u32x4 *d)
{
    const u32x4 *q = s + (size_t) blockIdx.x * 4096 + threadIdx.x;
    u32x4 *o = d + (size_t) blockIdx.x * 1024 + threadIdx.x;
    u32x4 v [16];
#pragma unroll
    for (int i = 0; i < 16; i++) v [i] = __builtin_nontemporal_load (q + 256 * i);
#pragma unroll
    for (int i = 0; i < 4; i++) __builtin_nontemporal_store (v [4 * i] ^ v [4 * i + 1] ^ v [4 * i + 2] ^ v [4 * i + 3], o + 256 * i);
}

// two read streams: 16 KiB from a and 16 KiB from b per workgroup
__global__ void __launch_bounds__ (256) k_read2 (const u32x4 *a, const u32x4 *b, uint32_t *sink)
{
    const u32x4 *p = a + (size_t) blockIdx.x * 1024 + threadIdx.x;
    const u32x4 *q = b + (size_t) blockIdx.x * 1024 + threadIdx.x;
    u32x4 v [8];
#pragma unroll
    for (int i = 0; i < 4; i++) { v [i] = __builtin_nontemporal_load (p + 256 * i); v [4 + i] = __builtin_nontemporal_load (q + 256 * i); }
    u32x4 x = v [0] ^ v [1] ^ v [2] ^ v [3] ^ v [4] ^ v [5] ^ v [6] ^ v [7];
    if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345679u) *sink = 1;
}

// mode 0: read, 1: write. bytes: a multiple of 16 KiB. Returns the mean of `iters' launches in microseconds.
extern "C" int
smol_zone_probe_pair (const void *src, void *dst, uint64_t dst_bytes, int reduce4, uint32_t warmup, uint32_t iters, double *us)
{
    static uint32_t *sink2 = nullptr;
    if (!sink2 && hipMalloc (&sink2, 4) != hipSuccess)
        return -1;
    hipEvent_t e0, e1;
    (void) hipEventCreate (&e0);
    (void) hipEventCreate (&e1);
    const uint32_t grid = (uint32_t) (dst_bytes >> 14);
    for (uint32_t i = 0; i < warmup + iters; i++)
    {
        if (i == warmup)
            (void) hipEventRecord (e0, 0);
        if (reduce4 == 2)
            hipLaunchKernelGGL (k_read2, dim3 (grid), dim3 (256), 0, 0, (const u32x4 *) src, (const u32x4 *) dst, sink2);
        else if (reduce4)
            hipLaunchKernelGGL (k_reduce4, dim3 (grid), dim3 (256), 0, 0, (const u32x4 *) src, (u32x4 *) dst);
        else
            hipLaunchKernelGGL (k_copy, dim3 (grid), dim3 (256), 0, 0, (const u32x4 *) src, (u32x4 *) dst);
    }
    (void) hipEventRecord (e1, 0);
    if (hipEventSynchronize (e1) != hipSuccess)
        return -2;
    float ms = 0;
    (void) hipEventElapsedTime (&ms, e0, e1);
    *us = (double) ms * 1e3 / iters;
    (void) hipEventDestroy (e0);
    (void) hipEventDestroy (e1);
    return 0;
}

extern "C" int
smol_zone_probe (void *base, uint64_t bytes, int mode, uint32_t warmup, uint32_t iters, double *us)
{
    static uint32_t *sink = nullptr;
    if (!sink && hipMalloc (&sink, 4) != hipSuccess)
        return -1;
    hipEvent_t e0, e1;
    (void) hipEventCreate (&e0);
    (void) hipEventCreate (&e1);
    const uint32_t grid = (uint32_t) (bytes >> 14);
    for (uint32_t i = 0; i < warmup + iters; i++)
    {
        if (i == warmup)
            (void) hipEventRecord (e0, 0);
        if (mode == 0)
            hipLaunchKernelGGL (k_read, dim3 (grid), dim3 (256), 0, 0, (const u32x4 *) base, sink);
        else
            hipLaunchKernelGGL (k_write, dim3 (grid), dim3 (256), 0, 0, (u32x4 *) base, i);
    }
    (void) hipEventRecord (e1, 0);
    if (hipEventSynchronize (e1) != hipSuccess)
        return -2;
    float ms = 0;
    (void) hipEventElapsedTime (&ms, e0, e1);
    *us = (double) ms * 1e3 / iters;
    (void) hipEventDestroy (e0);
    (void) hipEventDestroy (e1);
    return 0;
}
