// valu_rate.hip -- issue cost of the vector instructions the resampling kernels are built from (gfx950).
// One kernel per instruction: a wave runs ITER iterations of 16 independent copies (no dependency between
// consecutive instructions; each copy depends on itself from the previous iteration), timed with s_memtime
// by lane 0. Printed: cycles per instruction per wave with 1, 2 and 4 waves per SIMD resident (launch of
// 256 CUs x {4, 8, 16} waves). Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e)); return 1; } } while (0)
constexpr int ITER = 2000;

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

#define KERNEL(NAME, ASMSTR)                                                                        \
__global__ void __launch_bounds__ (1024) k_##NAME (uint32_t *out, uint64_t *cyc, uint32_t seed)        \
{                                                                                                   \
    uint32_t v [16], a = seed * 3u + threadIdx.x, b = seed | 1u, c = seed ^ 0x1234567u;             \
    for (int i = 0; i < 16; i++) v [i] = seed + i * 77u + threadIdx.x;                              \
    uint64_t t0 = __builtin_readcyclecounter ();                                                    \
    for (int it = 0; it < ITER; it++)                                                               \
    {                                                                                               \
        _Pragma ("unroll") for (int i = 0; i < 16; i++)                                             \
            asm volatile (ASMSTR : "+v" (v [i]) : "v" (a), "v" (b), "v" (c) : "vcc", "s10", "s11", "s12");                       \
    }                                                                                               \
    uint64_t t1 = __builtin_readcyclecounter ();                                                    \
    uint32_t r = 0;                                                                                 \
    for (int i = 0; i < 16; i++) r ^= v [i];                                                        \
    out [blockIdx.x * blockDim.x + threadIdx.x] = r;                                                \
    if ((threadIdx.x & 63) == 0) cyc [blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
}

KERNEL (v_add_u32,        "v_add_u32 %0, %0, %1")
KERNEL (v_and_b32,        "v_and_b32 %0, %0, %1")
KERNEL (v_perm_b32,       "v_perm_b32 %0, %0, %1, %2")
KERNEL (v_alignbyte,      "v_alignbyte_b32 %0, %0, %1, %2")
KERNEL (v_pk_add_u16,     "v_pk_add_u16 %0, %0, %1")
KERNEL (v_pk_mul_lo_u16,  "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL (v_pk_mad_u16,     "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL (v_pk_lshrrev_b16, "v_pk_lshrrev_b16 %0, 8, %0 op_sel_hi:[0,1]")
KERNEL (v_mul_u32_u24,    "v_mul_u32_u24 %0, %0, %1")
KERNEL (v_mad_u32_u24,    "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL (v_mul_lo_u32,     "v_mul_lo_u32 %0, %0, %1")
KERNEL (v_mad_u64_u32_lo, "v_mul_hi_u32 %0, %0, %1")
KERNEL (v_dot4_u32_u8,    "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL (v_dot2_u32_u16,   "v_dot2_u32_u16 %0, %0, %1, %2")
KERNEL (v_lshl_or_b32,    "v_lshl_or_b32 %0, %0, 8, %1")
KERNEL (v_and_or_b32,     "v_and_or_b32 %0, %0, %1, %2")
KERNEL (v_bfe_u32,        "v_bfe_u32 %0, %0, 8, 8")
KERNEL (v_lshrrev_b32,    "v_lshrrev_b32 %0, 3, %0")
KERNEL (v_cndmask,        "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL (v_cndmask_s,      "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL (cmp_cnd_vcc,      "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL (cmp_cnd_sgpr,     "v_cmp_lt_u32 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[10:11]")
KERNEL (cnd_vcc_e64,      "v_cndmask_b32_e64 %0, %0, %1, vcc")
KERNEL (v_mov_b32,        "v_mov_b32 %0, %1")
KERNEL (v_or_b32,         "v_or_b32 %0, %0, %1")
KERNEL (v_lshl_add_u32,   "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL (v_add3_u32,       "v_add3_u32 %0, %0, %1, %2")
KERNEL (v_pk_sub_u16,     "v_pk_sub_u16 %0, %0, %1")
KERNEL (v_cmp_lt_u32,     "v_cmp_lt_u32 vcc, %0, %1")
KERNEL (v_readfirstlane,  "v_readfirstlane_b32 s12, %0")
KERNEL (v_readlane,       "v_readlane_b32 s12, %0, 7")
KERNEL (v_bfi_b32,        "v_bfi_b32 %0, %1, %0, %2")
KERNEL (v_lshlrev_b32,    "v_lshlrev_b32 %0, 3, %0")
KERNEL (v_xor_b32,        "v_xor_b32 %0, %0, %1")
KERNEL (v_sub_u32,        "v_sub_u32 %0, %0, %1")
KERNEL (v_max_u32,        "v_max_u32 %0, %0, %1")
KERNEL (v_add_u16,        "v_add_u16 %0, %0, %1")
KERNEL (v_mad_u16,        "v_mad_u16 %0, %0, %1, %2")
KERNEL (v_mul_lo_u16,     "v_mul_lo_u16 %0, %0, %1")
KERNEL (v_mad_u32_u16,    "v_mad_u32_u16 %0, %0, %1, %2")
KERNEL (v_lshrrev_sdwa,   "v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
KERNEL (v_and_sdwa_b1,    "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL (v_add_sdwa_w1,    "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
KERNEL (v_mul_f32,        "v_mul_f32 %0, %0, %1")
KERNEL (v_cvt_u32_f32,    "v_cvt_u32_f32 %0, %0")
KERNEL (v_mov_dpp_shr1,   "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL (v_add_dpp_shr1,   "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL (v_sad_u8,         "v_sad_u8 %0, %0, %1, %2")
KERNEL (v_mul_sdwa_b0,    "v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")
KERNEL (v_pk_mul_sgpr,    "v_pk_mul_lo_u16 %0, s4, %0 op_sel_hi:[0,1]")
KERNEL (v_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
KERNEL (v_fma_f32,        "v_fma_f32 %0, %0, %1, %2")
KERNEL (v_pk_fma_f16,     "v_pk_fma_f16 %0, %0, %1, %2")
KERNEL (v_pk_mul_f16,     "v_pk_mul_f16 %0, %0, %1")
KERNEL (v_cvt_pk_u8_f32,  "v_cvt_pk_u8_f32 %0, %0, 1, %1")

// scalar unit: 16 independent s_add per iteration, all waves of the CU hammering the one scalar ALU
__global__ void __launch_bounds__ (1024) k_s_add (uint32_t *out, uint64_t *cyc, uint32_t seed)
{
    uint32_t s [16];
    for (int i = 0; i < 16; i++) s [i] = __builtin_amdgcn_readfirstlane (seed + i);
    uint64_t t0 = __builtin_readcyclecounter ();
    for (int it = 0; it < ITER; it++)
    {
#pragma unroll
        for (int i = 0; i < 16; i++)
            asm volatile ("s_add_i32 %0, %0, 7" : "+s" (s [i]));
    }
    uint64_t t1 = __builtin_readcyclecounter ();
    uint32_t r = 0;
    for (int i = 0; i < 16; i++) r ^= s [i];
    out [blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc [blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
// mixed: 8 VALU + 8 SALU per iteration in one wave (can one wave's scalar and vector work overlap? no -- in order;
// but two waves on one SIMD can: compare with the pure kernels)
__global__ void __launch_bounds__ (1024) k_mixed (uint32_t *out, uint64_t *cyc, uint32_t seed)
{
    uint32_t s [8], v [8], a = seed + threadIdx.x;
    for (int i = 0; i < 8; i++) { s [i] = __builtin_amdgcn_readfirstlane (seed + i); v [i] = seed * i + threadIdx.x; }
    uint64_t t0 = __builtin_readcyclecounter ();
    for (int it = 0; it < ITER; it++)
    {
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            asm volatile ("s_add_i32 %0, %0, 7" : "+s" (s [i]));
            asm volatile ("v_add_u32 %0, %0, %1" : "+v" (v [i]) : "v" (a));
        }
    }
    uint64_t t1 = __builtin_readcyclecounter ();
    uint32_t r = 0;
    for (int i = 0; i < 8; i++) r ^= s [i] ^ v [i];
    out [blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc [blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
// LDS: 16 independent ds_read_b32 / b64 / b128 per iteration at stride 4 / 8 / 16 bytes per lane (conflict-free)
template <int W>
__global__ void __launch_bounds__ (1024) k_lds (uint32_t *out, uint64_t *cyc, uint32_t seed)
{
    __shared__ __attribute__ ((aligned (16))) uint32_t lds [16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds [i] = seed + i;
    __syncthreads ();
    const uint32_t base = (threadIdx.x & 63) * W * 4 + (threadIdx.x >> 6) * 1024 * 4;
    uint32_t r = 0;
    uint64_t t0 = __builtin_readcyclecounter ();
    for (int it = 0; it < ITER / 4; it++)
    {
        uint32_t x [16][4];
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            const uint32_t ad = (base + i * 64 * W * 4) & 65535u;
            if (W == 1) asm volatile ("ds_read_b32 %0, %1" : "=v" (x [i][0]) : "v" (ad));
            if (W == 2) asm volatile ("ds_read_b64 %0, %1" : "=v" (*(uint64_t *) x [i]) : "v" (ad));
            if (W == 4) asm volatile ("ds_read_b128 %0, %1" : "=v" (*(__uint128_t *) x [i]) : "v" (ad));
        }
        asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i++) r ^= x [i][0];
    }
    uint64_t t1 = __builtin_readcyclecounter ();
    out [blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc [blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) * 4;   // ITER/4 iterations: scale to ITER
}

int main ()
{
    uint32_t *out; uint64_t *cyc;
    CK (hipMalloc (&out, 256 * 1024 * 4)); CK (hipMalloc (&cyc, 256 * 16 * 8));
    std::vector<uint64_t> h (256 * 16);
    printf ("%-22s %10s %10s %10s   (s_memtime ticks per instruction per wave; 100 MHz counter -> multiply by clock/100MHz)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    auto run = [&] (const char *name, auto kern) {
        printf ("%-22s", name);
        for (int wps = 1; wps <= 4; wps *= 2)
        {
            const int waves = 4 * wps;
            hipLaunchKernelGGL (kern, dim3 (256), dim3 (64 * waves), 0, 0, out, cyc, 12345u);
            hipLaunchKernelGGL (kern, dim3 (256), dim3 (64 * waves), 0, 0, out, cyc, 12345u);
            hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
            hipEventRecord (e0, 0);
            hipLaunchKernelGGL (kern, dim3 (256), dim3 (64 * waves), 0, 0, out, cyc, 12345u);
            hipEventRecord (e1, 0); hipEventSynchronize (e1);
            float ms = 0; (void) hipEventElapsedTime (&ms, e0, e1);
            { hipError_t le = hipGetLastError (); if (le != hipSuccess) printf (" [%s]", hipGetErrorString (le)); }
            // wall time per instruction per wave in ns (whole launch / (ITER * 16)): independent of the counter's unit
            printf (" %8.3f ns", ms * 1e6 / (ITER * 16.0));
            hipEventDestroy (e0); hipEventDestroy (e1);
        }
        printf ("\n");
    };
#define RUN(NAME) run (#NAME, k_##NAME);
    RUN (v_add_u32) RUN (v_and_b32) RUN (v_perm_b32) RUN (v_alignbyte) RUN (v_pk_add_u16) RUN (v_pk_mul_lo_u16) RUN (v_pk_mad_u16)
    RUN (v_pk_lshrrev_b16) RUN (v_mul_u32_u24) RUN (v_mad_u32_u24) RUN (v_mul_lo_u32) RUN (v_mad_u64_u32_lo) RUN (v_dot4_u32_u8)
    RUN (v_dot2_u32_u16) RUN (v_lshl_or_b32) RUN (v_and_or_b32) RUN (v_bfe_u32) RUN (v_lshrrev_b32) RUN (v_cndmask)
    RUN (v_mov_dpp_shr1) RUN (v_add_dpp_shr1) RUN (v_sad_u8) RUN (v_mul_sdwa_b0) RUN (v_pk_mul_sgpr)
    RUN (v_cvt_f32_ubyte0) RUN (v_fma_f32) RUN (v_pk_fma_f16) RUN (v_pk_mul_f16) RUN (v_cvt_pk_u8_f32)
    RUN (v_cndmask_s) RUN (cmp_cnd_vcc) RUN (cmp_cnd_sgpr) RUN (cnd_vcc_e64) RUN (v_mov_b32) RUN (v_or_b32) RUN (v_lshl_add_u32) RUN (v_add3_u32) RUN (v_pk_sub_u16) RUN (v_cmp_lt_u32)
    RUN (v_readfirstlane) RUN (v_readlane) RUN (v_bfi_b32) RUN (v_lshlrev_b32) RUN (v_xor_b32) RUN (v_sub_u32) RUN (v_max_u32)
    RUN (v_add_u16) RUN (v_mad_u16) RUN (v_mul_lo_u16) RUN (v_mad_u32_u16) RUN (v_lshrrev_sdwa) RUN (v_and_sdwa_b1) RUN (v_add_sdwa_w1)
    RUN (v_mul_f32) RUN (v_cvt_u32_f32)
    RUN (s_add) RUN (mixed)
    run ("ds_read_b32", k_lds<1>); run ("ds_read_b64", k_lds<2>); run ("ds_read_b128", k_lds<4>);
    printf ("(a full-rate vector instruction at 2.4 GHz issues every 4 cycles = 1.67 ns per wave at 1 wave per SIMD; with 2 and 4 waves per SIMD\n"
            " the launch takes 2x and 4x as long when the pipe is the limit, so the per-wave figure doubles)\n");
    return 0;
}
