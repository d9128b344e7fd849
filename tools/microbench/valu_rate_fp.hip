// valu_rate_fp.hip -- issue cost of the float instructions a fp32 formulation of the 11-bit taps would be built from
// (gfx950), same method as valu_rate.hip: 16 independent copies per iteration, wall time per instruction per wave with
// 1, 2 and 4 waves per SIMD. Round 5: can (p * F + q * (256 - F)) >> 8 on 11-bit values run as mul + fma + floor in the
// full-rate class instead of six packed 16-bit instructions per two channels?
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate_fp.hip -o valu_rate_fp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e)); return 1; } } while (0)
constexpr int ITER = 2000;

#define KERNEL(NAME, ASMSTR)                                                                        \
__global__ void __launch_bounds__ (1024) k_##NAME (uint32_t *out, uint32_t seed)                    \
{                                                                                                   \
    uint32_t v [16], a = seed * 3u + threadIdx.x, b = seed | 1u, c = seed ^ 0x1234567u;             \
    for (int i = 0; i < 16; i++) v [i] = seed + i * 77u + threadIdx.x;                              \
    for (int it = 0; it < ITER; it++)                                                               \
    {                                                                                               \
        _Pragma ("unroll") for (int i = 0; i < 16; i++)                                             \
            asm volatile (ASMSTR : "+v" (v [i]) : "v" (a), "v" (b), "v" (c) : "vcc");               \
    }                                                                                               \
    uint32_t r = 0;                                                                                 \
    for (int i = 0; i < 16; i++) r ^= v [i];                                                        \
    out [blockIdx.x * blockDim.x + threadIdx.x] = r;                                                \
}
// 64-bit destination (packed f32 pairs)
#define KERNEL64(NAME, ASMSTR)                                                                      \
__global__ void __launch_bounds__ (1024) k_##NAME (uint32_t *out, uint32_t seed)                    \
{                                                                                                   \
    uint64_t v [16], a = seed * 3u + threadIdx.x, b = seed | 1u, c = seed ^ 0x1234567u;             \
    for (int i = 0; i < 16; i++) v [i] = seed + i * 77u + threadIdx.x;                              \
    for (int it = 0; it < ITER; it++)                                                               \
    {                                                                                               \
        _Pragma ("unroll") for (int i = 0; i < 16; i++)                                             \
            asm volatile (ASMSTR : "+v" (v [i]) : "v" (a), "v" (b), "v" (c) : "vcc");               \
    }                                                                                               \
    uint64_t r = 0;                                                                                 \
    for (int i = 0; i < 16; i++) r ^= v [i];                                                        \
    out [blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t) (r ^ (r >> 32));                       \
}

KERNEL (v_add_u32,        "v_add_u32 %0, %0, %1")
KERNEL (v_mul_u32_u24,    "v_mul_u32_u24 %0, %0, %1")
KERNEL (v_mul_f32,        "v_mul_f32 %0, %0, %1")
KERNEL (v_add_f32,        "v_add_f32 %0, %0, %1")
KERNEL (v_fma_f32,        "v_fma_f32 %0, %0, %1, %2")
KERNEL (v_fmac_f32,       "v_fmac_f32 %0, %1, %2")
KERNEL (v_floor_f32,      "v_floor_f32 %0, %0")
KERNEL (v_trunc_f32,      "v_trunc_f32 %0, %0")
KERNEL (v_rndne_f32,      "v_rndne_f32 %0, %0")
KERNEL (v_fract_f32,      "v_fract_f32 %0, %0")
KERNEL (v_cvt_f32_u32,    "v_cvt_f32_u32 %0, %0")
KERNEL (v_cvt_u32_f32,    "v_cvt_u32_f32 %0, %0")
KERNEL (v_cvt_f32_f16,    "v_cvt_f32_f16 %0, %0")
KERNEL (v_cvt_f32_f16_hi, "v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
KERNEL (v_cvt_f16_f32,    "v_cvt_f16_f32 %0, %0")
KERNEL (v_cvt_pkrtz,      "v_cvt_pkrtz_f16_f32 %0, %0, %1")
KERNEL (v_fma_mix_lo,     "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]")              /* src0 = f16 low half of %0, others f32 */
KERNEL (v_fma_mix_hi,     "v_fma_mix_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]")   /* src0 = f16 high half */
KERNEL (v_fma_mix_2h,     "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]")              /* two f16 sources */
KERNEL (v_mad_mixlo_f16,  "v_fma_mixlo_f16 %0, %0, %1, %2")
KERNEL (v_dot2_f32_f16,   "v_dot2_f32_f16 %0, %0, %1, %2")
KERNEL (v_dot2c_f32_f16,  "v_dot2c_f32_f16 %0, %1, %2")
KERNEL (v_pk_fma_f16,     "v_pk_fma_f16 %0, %0, %1, %2")
KERNEL (v_med3_f32,       "v_med3_f32 %0, %0, %1, %2")
KERNEL (v_mul_legacy,     "v_mul_legacy_f32 %0, %0, %1")
KERNEL (v_lshrrev_b32,    "v_lshrrev_b32 %0, 8, %0")
KERNEL (v_lshl_sdwa_b1,   "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
KERNEL (v_mov_dpp_wshr1,  "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL64 (v_pk_fma_f32,   "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL64 (v_pk_mul_f32,   "v_pk_mul_f32 %0, %0, %1")
KERNEL64 (v_pk_add_f32,   "v_pk_add_f32 %0, %0, %1")
KERNEL64 (v_pk_mov_b32,   "v_pk_mov_b32 %0, %0, %1")
KERNEL64 (v_lshrrev_b64,  "v_lshrrev_b64 %0, 8, %0")

// LDS look-ups: 16 independent reads per iteration, per-lane pseudo-random indices into a 256-entry u16 table (the from_srgb /
// opaque_lin pattern), plain and as d16 / d16_hi pairs filling both halves of one register
template <int MODE>
__global__ void __launch_bounds__ (1024) k_lut (uint32_t *out, uint32_t seed)
{
    __shared__ __attribute__ ((aligned (16))) uint16_t tab [4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab [i] = (uint16_t) (seed + i * 13);
    __syncthreads ();
    uint32_t idx [16], r = 0;
    for (int i = 0; i < 16; i++) idx [i] = (((threadIdx.x * 2654435761u + i * 40503u + seed) >> 13) & 255u) * 2u;
    for (int it = 0; it < ITER / 4; it++)
    {
        uint32_t x [16];
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            x [i] = r;
            if (MODE == 0) asm volatile ("ds_read_u16 %0, %1" : "=v" (x [i]) : "v" (idx [i]));
            if (MODE == 1 && !(i & 1)) asm volatile ("ds_read_u16_d16 %0, %1" : "+v" (x [i]) : "v" (idx [i]));
            if (MODE == 1 && (i & 1)) asm volatile ("ds_read_u16_d16_hi %0, %1" : "+v" (x [i - 1]) : "v" (idx [i]));
            if (MODE == 2) asm volatile ("ds_read_b32 %0, %1" : "=v" (x [i]) : "v" (idx [i] * 2u));
        }
        asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i++) { r ^= x [i]; idx [i] = (idx [i] * 5u + 2u + (r & 2u)) & 510u; }
    }
    out [blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main ()
{
    uint32_t *out;
    CK (hipMalloc (&out, 256 * 1024 * 4));
    printf ("%-22s %10s %10s %10s   (ns per instruction per wave-slot: launch time / (%d x 16))\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", ITER);
    auto run = [&] (const char *name, auto kern, double scale) {
        printf ("%-22s", name);
        for (int wps = 1; wps <= 4; wps *= 2)
        {
            const int waves = 4 * wps;
            hipLaunchKernelGGL (kern, dim3 (256), dim3 (64 * waves), 0, 0, out, 12345u);
            hipLaunchKernelGGL (kern, dim3 (256), dim3 (64 * waves), 0, 0, out, 12345u);
            hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
            hipEventRecord (e0, 0);
            hipLaunchKernelGGL (kern, dim3 (256), dim3 (64 * waves), 0, 0, out, 12345u);
            hipEventRecord (e1, 0); hipEventSynchronize (e1);
            float ms = 0; (void) hipEventElapsedTime (&ms, e0, e1);
            { hipError_t le = hipGetLastError (); if (le != hipSuccess) printf (" [%s]", hipGetErrorString (le)); }
            printf (" %8.3f ns", ms * 1e6 * scale / (ITER * 16.0));
            hipEventDestroy (e0); hipEventDestroy (e1);
        }
        printf ("\n");
    };
#define RUN(NAME) run (#NAME, k_##NAME, 1.0);
    RUN (v_add_u32) RUN (v_mul_u32_u24) RUN (v_mul_f32) RUN (v_add_f32) RUN (v_fma_f32) RUN (v_fmac_f32) RUN (v_floor_f32) RUN (v_trunc_f32)
    RUN (v_rndne_f32) RUN (v_fract_f32) RUN (v_cvt_f32_u32) RUN (v_cvt_u32_f32) RUN (v_cvt_f32_f16) RUN (v_cvt_f32_f16_hi) RUN (v_cvt_f16_f32)
    RUN (v_cvt_pkrtz) RUN (v_fma_mix_lo) RUN (v_fma_mix_hi) RUN (v_fma_mix_2h) RUN (v_mad_mixlo_f16) RUN (v_dot2_f32_f16) RUN (v_dot2c_f32_f16)
    RUN (v_pk_fma_f16) RUN (v_med3_f32) RUN (v_mul_legacy) RUN (v_lshrrev_b32) RUN (v_lshl_sdwa_b1) RUN (v_mov_dpp_wshr1)
    RUN (v_pk_fma_f32) RUN (v_pk_mul_f32) RUN (v_pk_add_f32) RUN (v_pk_mov_b32) RUN (v_lshrrev_b64)
    run ("lut ds_read_u16", k_lut<0>, 4.0); run ("lut d16 + d16_hi", k_lut<1>, 4.0); run ("lut ds_read_b32", k_lut<2>, 4.0);
    return 0;
}
