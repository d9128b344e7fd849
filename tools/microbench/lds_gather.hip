// What does a per-lane table look-up (LDS gather at random indices) cost on gfx950, and what does replicating the
// table per bank buy? The sRGB-linear conversions do 4-7 such look-ups per pixel (smol_kernels.hip.h LutLds).
//   hipcc --offload-arch=gfx950 -O3 -o lds_gather lds_gather.hip && ./lds_gather
// Full chip: 512 workgroups x 1024 threads, every lane 8 independent look-ups per iteration; addresses are random per
// lane and re-randomised per iteration with one v_xor (the VALU cost per look-up stays below the LDS cost).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

enum { V_U16 = 0, V_B32 = 1, V_U8_2K = 2, V_REP32 = 3, V_REP16 = 4, V_REP32_U16 = 5, V_REP8 = 6, V_SAME = 7, V_LINEAR = 8, NV };
static const char *names [NV] = {
    "u16 [256], random index            (from_srgb today)",
    "u32 [256], random index            (inverse tables today)",
    "u8 [2048], random index            (to_srgb today)",
    "u32 [256] x 32 copies, own bank    (idx * 128 + (lane & 31) * 4)",
    "u32 [256] x 16 copies, 2 banks each",
    "u16 [256] x 32 copies (pairs), own bank",
    "u32 [256] x 8 copies, 4 banks each",
    "u32, every lane the same index     (broadcast)",
    "u32, lane-linear index             (conflict free)",
};

template <int V>
__global__ void __launch_bounds__ (1024) k (uint32_t *out, int iters, uint32_t seed)
{
    extern __shared__ __attribute__ ((aligned (16))) uint8_t lds [];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t i = tid; i < 65536 / 4; i += 1024)
        reinterpret_cast<uint32_t *> (lds) [i] = i * 2654435761u;
    __syncthreads ();
    uint32_t s = seed ^ (blockIdx.x * 1024u + tid) * 747796405u;
    uint32_t addr [8], mask;
#pragma unroll
    for (int k8 = 0; k8 < 8; k8++)
    {
        s = s * 1664525u + 1013904223u;
        const uint32_t idx = (s >> 11);
        switch (V)
        {
            case V_U16:       addr [k8] = (idx & 255u) * 2u; mask = 255u * 2u; break;
            case V_B32:       addr [k8] = (idx & 255u) * 4u; mask = 255u * 4u; break;
            case V_U8_2K:     addr [k8] = idx & 2047u; mask = 2047u; break;
            case V_REP32:     addr [k8] = (idx & 255u) * 128u + (lane & 31u) * 4u; mask = 255u * 128u; break;
            case V_REP16:     addr [k8] = ((idx & 255u) >> 1) * 128u + (lane & 15u) * 8u + (idx & 1u) * 4u; mask = 127u * 128u + 4u; break;
            case V_REP32_U16: addr [k8] = ((idx & 255u) >> 1) * 128u + (lane & 31u) * 4u + (idx & 1u) * 2u; mask = 127u * 128u + 2u; break;
            case V_REP8:      addr [k8] = ((idx & 255u) >> 2) * 128u + (lane & 7u) * 16u + (idx & 3u) * 4u; mask = 63u * 128u + 12u; break;
            case V_SAME:      addr [k8] = (k8 * 52u) & 1020u; mask = 0; break;
            default:          addr [k8] = ((lane + k8 * 64u) * 4u) & 4095u; mask = 0; break;
        }
    }
    uint32_t acc = 0, r = seed;
    for (int it = 0; it < iters; it++)
    {
        r = r * 1664525u + 1013904223u;                 // wave-uniform (scalar unit)
        const uint32_t x = (r >> 9) & mask;
        uint32_t v [8];
#pragma unroll
        for (int k8 = 0; k8 < 8; k8++)
        {
            addr [k8] ^= x;
            if (V == V_U16 || V == V_REP32_U16)
                asm volatile ("ds_read_u16 %0, %1" : "=v" (v [k8]) : "v" (addr [k8]) : "memory");
            else if (V == V_U8_2K)
                asm volatile ("ds_read_u8 %0, %1" : "=v" (v [k8]) : "v" (addr [k8]) : "memory");
            else
                asm volatile ("ds_read_b32 %0, %1" : "=v" (v [k8]) : "v" (addr [k8]) : "memory");
        }
        asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k8 = 0; k8 < 8; k8++)
            acc ^= v [k8];
    }
    out [blockIdx.x * 1024u + tid] = acc;
}

template <int V>
static void run (uint32_t *out, int iters)
{
    const int wgs = 512;
    hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
    hipFuncSetAttribute (reinterpret_cast<const void *> (k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL (k<V>, dim3 (wgs), dim3 (1024), 65536, 0, out, 10, 1u);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++)
    {
        hipEventRecord (e0);
        hipLaunchKernelGGL (k<V>, dim3 (wgs), dim3 (1024), 65536, 0, out, iters, 7u + rep);
        hipEventRecord (e1); hipEventSynchronize (e1);
        float ms; hipEventElapsedTime (&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // wave-level look-ups per CU: 2 workgroups x 16 waves x iters x 8
    const double per_cu = 2.0 * 16 * iters * 8.0;
    printf ("%-66s %8.1f us  %6.2f ns per wave look-up per CU = %5.1f cycles at 2.4 GHz\n", names [V], best * 1e3,
            best * 1e6 / per_cu, best * 1e6 / per_cu * 2.4);
}

int main ()
{
    uint32_t *out; hipMalloc (&out, 512 * 1024 * 4);
    const int iters = 2000;
    run<V_U16> (out, iters); run<V_B32> (out, iters); run<V_U8_2K> (out, iters); run<V_REP32> (out, iters);
    run<V_REP16> (out, iters); run<V_REP32_U16> (out, iters); run<V_REP8> (out, iters); run<V_SAME> (out, iters);
    run<V_LINEAR> (out, iters);
    return 0;
}
