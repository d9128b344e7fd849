/* smol_k_probe.hip -- the one kernel that is not part of the scaling path: two streaming READS side by side (non-temporal,
 * 16 KiB of each per workgroup), which smol_hip_memory_zones() times over pairs of windows of a caller's region. */

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smol_launch.h"

namespace smolhip {

typedef uint32_t probe_u32x4 __attribute__ ((ext_vector_type (4)));

__global__ void __launch_bounds__ (256)
k_zone_read2 (const probe_u32x4 *a, const probe_u32x4 *b)
{
    const probe_u32x4 *p = a + (size_t) blockIdx.x * 1024 + threadIdx.x;
    const probe_u32x4 *q = b + (size_t) blockIdx.x * 1024 + threadIdx.x;
    probe_u32x4 v [8];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        v [i] = __builtin_nontemporal_load (p + 256 * i);
        v [4 + i] = __builtin_nontemporal_load (q + 256 * i);
    }
    /* nothing is written: the empty statements keep the loads alive */
#pragma unroll
    for (int i = 0; i < 8; i++)
        asm volatile ("" : : "v" (v [i]));
}

namespace launch {

void
zone_read2 (const void *a, const void *b, uint64_t bytes_each, hipStream_t s)
{
    hipLaunchKernelGGL (k_zone_read2, dim3 ((uint32_t) (bytes_each >> 14)), dim3 (256), 0, s,
                        static_cast<const probe_u32x4 *> (a), static_cast<const probe_u32x4 *> (b));
}

} /* namespace launch */
} /* namespace smolhip */
