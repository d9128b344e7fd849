// tlb_probe: the latency of a chain of dependent reads over an existing device allocation at a given stride, as a shared
// library for tools/bench_configs.py (SMOL_BENCH_ARENA_PROBE=tools/microbench/libtlb_probe.so): what does the address
// translation of THIS allocation cost? A chain at a stride of 64 KiB stays inside one page-table fragment for many steps
// when the driver mapped the allocation with large fragments, and leaves it at every step when it did not; a chain at a
// stride of 2 MiB + leaves any fragment up to 2 MiB at every step (the "always walks" reference).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/microbench/libtlb_probe.so tools/microbench/tlb_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void chase (const uint8_t *base, uint64_t bytes, uint64_t stride, uint32_t n, uint32_t *sink)
{
    uint64_t off = 0;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < n; i++)
    {
        uint32_t v = *reinterpret_cast<const volatile uint32_t *> (base + off);
        acc += v;
        uint32_t z = v;
        asm volatile ("v_and_b32 %0, 0, %0" : "+v" (z));       // 0 at run time, unknown to the compiler: the next address waits for the load
        off += stride + z;
        if (off + 4 > bytes)
            off -= bytes - 64;                                      // wrap, shifted by a cache line
        if (off + 4 > bytes)
            off = 0;
    }
    *sink = acc;
}

extern "C" int
smol_probe_latency (const void *base, uint64_t bytes, uint64_t stride, uint32_t n, double *ns_per_access)
{
    uint32_t *sink = nullptr;
    hipEvent_t e0, e1;
    if (hipMalloc (&sink, 4) != hipSuccess)
        return -1;
    hipEventCreate (&e0);
    hipEventCreate (&e1);
    hipLaunchKernelGGL (chase, dim3 (1), dim3 (1), 0, 0, (const uint8_t *) base, bytes, stride, n / 8 + 1, sink);   // warm the clocks
    hipEventRecord (e0, 0);
    hipLaunchKernelGGL (chase, dim3 (1), dim3 (1), 0, 0, (const uint8_t *) base, bytes, stride, n, sink);
    hipEventRecord (e1, 0);
    if (hipEventSynchronize (e1) != hipSuccess)
        return -2;
    float ms = 0;
    hipEventElapsedTime (&ms, e0, e1);
    *ns_per_access = (double) ms * 1e6 / n;
    hipEventDestroy (e0);
    hipEventDestroy (e1);
    hipFree (sink);
    return 0;
}
