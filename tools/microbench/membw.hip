// membw.hip -- what HBM gives a pure streaming kernel on this box: read-only, write-only
// (plain and non-temporal), and copy, for the buffer sizes of the BASELINE configs.
// Calibration for the roofline fractions in DESIGN.md. Build: hipcc -O3 --offload-arch=gfx950 membw.hip -o membw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e)); return 1; } } while (0)

template <bool NT> __global__ void __launch_bounds__ (256) k_write (u32x4 *p, size_t n, uint32_t v)
{
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    const u32x4 x = { v, v + 1, v + 2, (uint32_t) i };
    if (i < n) { if (NT) __builtin_nontemporal_store (x, p + i); else p [i] = x; }
}
template <bool NT> __global__ void __launch_bounds__ (256) k_read (const u32x4 *p, size_t n, uint32_t *sink)
{
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32x4 x = NT ? __builtin_nontemporal_load (p + i) : p [i];
    if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345679u) *sink = 1;
}
template <bool NT> __global__ void __launch_bounds__ (256) k_copy (const u32x4 *s, u32x4 *d, size_t n)
{
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (NT) __builtin_nontemporal_store (__builtin_nontemporal_load (s + i), d + i); else d [i] = s [i];
}
// each thread writes 4 consecutive vectors of a "row" (like a magnify kernel: 64 B per lane)
__global__ void __launch_bounds__ (256) k_write4 (u32x4 *p, size_t n, uint32_t v)
{
    size_t i = ((size_t) blockIdx.x * 256 + threadIdx.x);
    const u32x4 x = { v, v + 1, v + 2, (uint32_t) i };
    size_t base = (i / 64) * 256 + (i % 64);
    for (int k = 0; k < 4; k++) if (base + 64 * k < n) __builtin_nontemporal_store (x, p + base + 64 * k);
}
int main ()
{
    const size_t bytes = (size_t) 1 << 30;            // 1 GiB: well past the 256 MB infinity cache
    const size_t n = bytes / 16;
    u32x4 *a, *b; uint32_t *sink;
    CK (hipMalloc (&a, bytes)); CK (hipMalloc (&b, bytes)); CK (hipMalloc (&sink, 4));
    CK (hipMemset (a, 1, bytes)); CK (hipMemset (b, 2, bytes));
    hipEvent_t e0, e1; CK (hipEventCreate (&e0)); CK (hipEventCreate (&e1));
    const unsigned grid = (unsigned) ((n + 255) / 256);
    auto run = [&] (const char *name, double traffic, auto launch) {
        for (int i = 0; i < 3; i++) launch ();
        hipEventRecord (e0, 0);
        const int it = 20;
        for (int i = 0; i < it; i++) launch ();
        hipEventRecord (e1, 0); hipEventSynchronize (e1);
        float ms; hipEventElapsedTime (&ms, e0, e1);
        printf ("%-28s %8.1f us  %7.1f GB/s  (%.1f %% of 8 TB/s)\n", name, ms * 1e3 / it, traffic / (ms * 1e-3 / it) / 1e9, traffic / (ms * 1e-3 / it) / 8e12 * 100);
    };
    run ("read 1 GiB", bytes, [&] { hipLaunchKernelGGL (k_read<false>, dim3 (grid), dim3 (256), 0, 0, a, n, sink); });
    run ("read 1 GiB nt", bytes, [&] { hipLaunchKernelGGL (k_read<true>, dim3 (grid), dim3 (256), 0, 0, a, n, sink); });
    run ("write 1 GiB", bytes, [&] { hipLaunchKernelGGL (k_write<false>, dim3 (grid), dim3 (256), 0, 0, a, n, 7u); });
    run ("write 1 GiB nt", bytes, [&] { hipLaunchKernelGGL (k_write<true>, dim3 (grid), dim3 (256), 0, 0, a, n, 7u); });
    run ("write 1 GiB nt 64B/lane", bytes, [&] { hipLaunchKernelGGL (k_write4, dim3 (grid / 4), dim3 (256), 0, 0, a, n, 7u); });
    run ("copy 1 GiB -> 1 GiB", 2.0 * bytes, [&] { hipLaunchKernelGGL (k_copy<false>, dim3 (grid), dim3 (256), 0, 0, a, b, n); });
    run ("copy 1 GiB -> 1 GiB nt", 2.0 * bytes, [&] { hipLaunchKernelGGL (k_copy<true>, dim3 (grid), dim3 (256), 0, 0, a, b, n); });
    run ("hipMemsetAsync 1 GiB", bytes, [&] { hipMemsetAsync (a, 3, bytes, 0); });
    run ("hipMemcpyAsync d2d 1 GiB", 2.0 * bytes, [&] { hipMemcpyAsync (b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
