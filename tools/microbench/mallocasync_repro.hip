// mallocasync_repro.hip -- does stream-ordered allocation hand a kernel stale data on this ROCm build?
// Round 1 saw 1048 of 4000 two-pass box jobs wrong when the intermediate image came from
// hipMallocAsync/hipFreeAsync (tools/fuzz/fuzz_box2_gpu.py) and none with hipMalloc. This isolates the
// pattern: producer kernel fills a scratch buffer, consumer kernel checks it, the scratch buffer comes
// from (a) hipMallocAsync on the same stream, freed with hipFreeAsync right after the consumer,
// (b) the same on one of four streams in rotation (the library's staging lanes), (c) hipMalloc once.
// Reports mismatching words per mode. Build: hipcc -O2 --offload-arch=gfx950 mallocasync_repro.hip -o mallocasync_repro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e)); return 1; } } while (0)

__global__ void produce (uint32_t *tmp, size_t n, uint32_t tag)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
        tmp [i] = tag ^ (uint32_t) i;
}

__global__ void consume (const uint32_t *tmp, size_t n, uint32_t tag, unsigned long long *bad)
{
    unsigned long long b = 0;
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
        b += tmp [i] != (tag ^ (uint32_t) i);
    if (b) atomicAdd (bad, b);
}

int main ()
{
    hipStream_t st [4];
    for (auto &s : st) CK (hipStreamCreateWithFlags (&s, hipStreamNonBlocking));
    unsigned long long *bad; CK (hipMalloc (&bad, 8));
    uint32_t *fixed; CK (hipMalloc (&fixed, 64u << 20));
    for (int mode = 0; mode < 3; mode++)
    {
        CK (hipMemset (bad, 0, 8));
        const int iters = 3000;
        for (int it = 0; it < iters; it++)
        {
            const size_t n = (size_t) (1 + (it * 7919) % 4000) * 1024;       // 4 KB .. 16 MB, varying
            hipStream_t s = (mode == 1) ? st [it & 3] : st [0];
            uint32_t *tmp = fixed;
            if (mode != 2) CK (hipMallocAsync ((void **) &tmp, n * 4, s));
            hipLaunchKernelGGL (produce, dim3 (256), dim3 (256), 0, s, tmp, n, 0x9e3779b9u * (uint32_t) it);
            hipLaunchKernelGGL (consume, dim3 (256), dim3 (256), 0, s, tmp, n, 0x9e3779b9u * (uint32_t) it, bad);
            if (mode != 2) CK (hipFreeAsync (tmp, s));
        }
        CK (hipDeviceSynchronize ());
        unsigned long long h; CK (hipMemcpy (&h, bad, 8, hipMemcpyDeviceToHost));
        printf ("%-52s %d iterations: %llu mismatching words\n",
                mode == 0 ? "hipMallocAsync/hipFreeAsync, one stream" : mode == 1 ? "hipMallocAsync/hipFreeAsync, four streams in rotation" : "one hipMalloc buffer, one stream",
                iters, h);
    }
    return 0;
}
