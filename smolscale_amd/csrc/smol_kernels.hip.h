/* smol_kernels.hip.h -- device code of the HIP backend (gfx950 / CDNA4 only).
 *
 * Included by the kernel translation units (smol_k_*.hip). Everything here is new code; the arithmetic it has
 * to reproduce is the reference's integer fixed-point path (citations per
 * function, paths relative to the reference tree).
 *
 * Kernel classes
 *   k_general<WIDE>   any job. One thread per output pixel; source pixels are
 *                     gathered straight from global memory. Correctness backstop
 *                     and the path for exotic jobs (sRGB-linear, premul16, box
 *                     with huge spans, ...).
 *   k_pow2<HX,HY>     exact 2:1 / 4:1 / 8:1 bilinear-with-halvings on 32-bit
 *                     premultiplied pixels (BASELINE configs 1, 2 and 5): every
 *                     weight is 128, so a tap is a per-byte floor average and
 *                     the whole 2^n x 2^m block reduces in packed registers.
 *                     Pure streaming: each lane issues 16-byte loads that are
 *                     contiguous across the wave (1 KiB per load instruction),
 *                     no LDS, no tables, every source byte read exactly once.
 *   k_pow2x<H,IN>     the same streaming shape for every other pixel-type pair
 *                     and for the 128bpp class (smol_kernel_pow2x.hip.h).
 *   k_up<IN,OUT>      bilinear magnification on both axes, organised around the
 *                     stores (smol_kernel_up.hip.h).
 *   k_lind<HX,IN>     bilinear minification on both axes for the 128bpp class
 *                     (smol_kernel_lind.hip.h).
 *   k_tile / k_wtile  LDS-tiled kernels for arbitrary ratios (smol_kernel_tile.hip.h,
 *                     smol_kernel_wtile.hip.h).
 */

#pragma once

#ifdef SMOL_KERNEL_EMU
/* tests/emu compiles these kernel bodies for the HOST (clang++, no GPU) to debug
 * kernel logic in the build container; never part of the shipped library. */
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#define SMOL_LUT_QUAL static __constant__     /* one copy per translation unit */
#include "smol_luts.h"

#include "smol_jobdev.h"

namespace smolhip {

/* Image pointers arrive inside a descriptor struct, so the compiler only knows
 * them as generic ("flat") pointers and would emit flat_load / flat_store. They
 * always point to device global memory: say so, and get global_load / global_store
 * (scalar base + 32-bit lane offset forms, vmcnt-only accounting). */
#ifdef SMOL_KERNEL_EMU
#define SMOL_AS_GLOBAL
#else
#define SMOL_AS_GLOBAL __attribute__ ((address_space (1)))
#endif
typedef SMOL_AS_GLOBAL uint8_t gbyte;

__device__ __forceinline__ gbyte *gptr (void *p)
{
    return reinterpret_cast<gbyte *> (reinterpret_cast<uintptr_t> (p));
}

__device__ __forceinline__ const gbyte *gptr (const void *p)
{
    return reinterpret_cast<const gbyte *> (reinterpret_cast<uintptr_t> (p));
}

/* "every active lane of the wave sees an opaque pixel": the sRGB-linear conversions have a two-look-up shortcut for
 * alpha = 255 whose results equal the general chain's. Taken per lane it makes a wave with ANY translucent pixel run
 * both sides of the branch; taken per wave, photographs (all opaque) run the short side only and everything else the
 * general side only. The host emulation runs lanes one after the other and decides per lane (same bytes either way). */
__device__ __forceinline__ bool smol_wave_all_opaque (uint32_t a)
{
#ifdef SMOL_KERNEL_EMU
    return a == 255u;
#else
    return __builtin_amdgcn_ballot_w64 (a != 255u) == 0;
#endif
}

/* Descriptor `idx' of a device-resident table (idx wave-uniform). Fetched as whole dwords: the
 * compiler then uses scalar loads (s_load_dwordxN through the scalar cache) and picks the 8- and
 * 16-bit fields out of SGPRs. A plain struct copy loads every sub-dword member with its own
 * global_load_ubyte / _ushort -- vector-memory latency at the head of every workgroup: config 4
 * ran 159 us that way against 149 us with the descriptor in the kernel arguments. */
struct JobDevRaw { uint32_t w [sizeof (JobDev) / 4]; };
static_assert (sizeof (JobDev) % 4 == 0 && sizeof (JobDevRaw) == sizeof (JobDev), "JobDev is copied dword-wise");

__device__ __forceinline__ JobDev load_job (const JobDev *jobs, uint32_t idx)
{
    JobDevRaw raw;
    const uint32_t *p = reinterpret_cast<const uint32_t *> (jobs + idx);
#pragma unroll
    for (uint32_t i = 0; i < sizeof (JobDev) / 4; i++)
        raw.w [i] = p [i];
#ifndef SMOL_KERNEL_EMU
    /* every dword has to exist whole in a scalar register (after all loads have been issued):
     * keeps the optimiser from narrowing a load to the bytes one field needs, which only vector
     * loads can fetch */
#pragma unroll
    for (uint32_t i = 0; i < sizeof (JobDev) / 4; i++)
        asm volatile ("" : "+s" (raw.w [i]));
#endif
    JobDev J;
    __builtin_memcpy (&J, &raw, sizeof (J));
    return J;
}

/* Output stores. Output pixels are written once and never read back by the kernel, so they should not
 * occupy the XCD's L2 on their way out: WRITE-THROUGH stores (the sc1 bit; what an agent-scope relaxed
 * atomic store compiles to) leave the line out of L2 (MI355X_MICROARCH.md, stores of each flavour) and
 * the input stream flows better -- the headline kernel at 32 frames: sc1 336 us, plain 346, non-temporal
 * 351 (tools/microbench/pow2_variants.hip, profiles/r02m_store_policy.txt). 4- and 8-byte forms are
 * compiler builtins; the 16-byte form has none and is one asm instruction (ending in s_nop 1: the
 * data registers of a >64-bit store must not be overwritten in the next wait state, cdna_hip_programming.md
 * 5.7; an uncounted store only ever makes the compiler's vmcnt waits stricter). */
__device__ __forceinline__ void store_out32 (SMOL_AS_GLOBAL uint32_t *p, uint32_t v)
{
#if defined (SMOL_KERNEL_EMU)
    *p = v;
#elif defined (SMOL_STORES_NT)
    __builtin_nontemporal_store (v, p);
#elif defined (SMOL_STORES_PLAIN)
    *p = v;
#else
    __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

__device__ __forceinline__ void store_out64 (SMOL_AS_GLOBAL uint64_t *p, uint32_t lo, uint32_t hi)
{
#if defined (SMOL_KERNEL_EMU)
    *p = (uint64_t) lo | ((uint64_t) hi << 32);
#elif defined (SMOL_STORES_NT)
    __builtin_nontemporal_store ((uint64_t) lo | ((uint64_t) hi << 32), p);
#elif defined (SMOL_STORES_PLAIN)
    *p = (uint64_t) lo | ((uint64_t) hi << 32);
#else
    __hip_atomic_store (p, (uint64_t) lo | ((uint64_t) hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

/* ------------------------------------------------------------------------- *
 * Per-channel arithmetic                                                    *
 * ------------------------------------------------------------------------- */

/* unassociated -> premultiplied 8 bit: smolscale-generic.c:238-244 */
__device__ __forceinline__ uint32_t premul8 (uint32_t c, uint32_t a)
{
    return ((__umul24 (c + 1u, a + 1u) - 1u) >> 8) & 0xffu;
}

/* linear 11 bit x alpha: smolscale-generic.c:261-269 */
__device__ __forceinline__ uint32_t premul8l (uint32_t l, uint32_t a)
{
    return ((__umul24 (l + 1u, (a << 3) + 1u) - 1u) >> 11) & 0x7ffu;
}

/* (value * inverse) >> shift & mask for the un-premultiply tables. The reference computes the product
 * in 64 bits; only bits below 32 of it are ever kept (shift + mask width <= 30), every lane value of
 * either storage class is below 2^24 and every table entry below 2^21 -- so the low 32 bits of a
 * 24 x 24-bit product (one full-rate v_mul_u32_u24 instead of a v_mul_lo / v_mul_hi pair) are exact.
 * Table maxima pinned by tests/test_golden.py. */
__device__ __forceinline__ uint32_t inv_mul (uint32_t v, uint32_t inv)
{
    return __umul24 (v, inv);
}

/* Where the six parity tables are read from. LutConst: the __constant__ copies
 * (vector loads through L1/L2 when the index differs per lane). LutLds: a
 * workgroup-private copy in LDS, for the 128bpp-class kernels whose unpack/pack
 * does several table lookups per pixel. */
struct LutConst
{
    __device__ __forceinline__ uint32_t inv_p8 (uint32_t a) const { return smol_inv_p8_lut [a]; }
    __device__ __forceinline__ uint32_t inv_p8l (uint32_t a) const { return smol_inv_p8l_lut [a]; }
    __device__ __forceinline__ uint32_t inv_p16 (uint32_t a) const { return smol_inv_p16_lut [a]; }
    __device__ __forceinline__ uint32_t inv_p16l (uint32_t a) const { return smol_inv_p16l_lut [a]; }
    __device__ __forceinline__ uint32_t from_srgb (uint32_t v) const { return smol_from_srgb_lut [v]; }
    __device__ __forceinline__ uint32_t to_srgb (uint32_t v) const { return smol_to_srgb_lut [v]; }
    /* opaque pixel (alpha = 255), sRGB byte -> premultiplied linear: the chain of unpack_word */
    __device__ __forceinline__ uint32_t opaque_lin (uint32_t v) const
    {
        const uint32_t u = (__umul24 (v, smol_inv_p8_lut [255]) >> 13) & 0xffu;
        return premul8l (smol_from_srgb_lut [u], 255u);
    }
};

constexpr uint32_t LUT_LDS_BYTES = SMOL_LUT_LDS_BYTES;

/* LUT_AT_LDS_0: the workgroup's table copy starts at LDS address 0 -- true of every kernel that keeps it at the head of
 * its dynamic LDS and has no static LDS (k_lind / k_linh / k_linv, k_upg, k_boxh / k_boxv; their entries trap if that ever
 * changes). The table pointers are then made from INTEGERS, and a look-up is `ds_read v, v_index offset:<table>'. Through
 * the `lds' symbol the compiler adds the base -- a literal 0 it only learns at link time -- to every index in a vector
 * instruction of its own: one more instruction per byte-table look-up (to_srgb: three per output pixel of every
 * gamma-correct job). */
struct LutAtLds0 {};
constexpr LutAtLds0 LUT_AT_LDS_0 = {};

#ifdef SMOL_KERNEL_EMU
#define SMOL_AS_LDS
#else
#define SMOL_AS_LDS __attribute__ ((address_space (3)))
#endif

struct LutLds
{
    const SMOL_AS_LDS uint32_t *p8, *p8l, *p16, *p16l;
    const SMOL_AS_LDS uint16_t *fs, *oq;
    const SMOL_AS_LDS uint8_t *ts;

    /* Typed LDS pointers: going through an untyped integer here once made the compiler forget that the tables live
     * in LDS (flat_load instead of ds_read: the gamma-correct rows ran 10-20 % slower for a few commits of round 3).
     * The 64bpp-class instantiations of the shared bodies have no table copy; they pass any valid LDS address. */
    __device__ __forceinline__ explicit LutLds (const uint8_t *base)
#ifdef SMOL_KERNEL_EMU
        : p8 (reinterpret_cast<const uint32_t *> (base)),
          p8l (reinterpret_cast<const uint32_t *> (base + 1024)),
          p16 (reinterpret_cast<const uint32_t *> (base + 2048)),
          p16l (reinterpret_cast<const uint32_t *> (base + 3072)),
          fs (reinterpret_cast<const uint16_t *> (base + 4096)),
          oq (reinterpret_cast<const uint16_t *> (base + 6656)),
          ts (base + 4608) {}
#else
        : p8 ((const SMOL_AS_LDS uint32_t *) (base)),
          p8l ((const SMOL_AS_LDS uint32_t *) (base + 1024)),
          p16 ((const SMOL_AS_LDS uint32_t *) (base + 2048)),
          p16l ((const SMOL_AS_LDS uint32_t *) (base + 3072)),
          fs ((const SMOL_AS_LDS uint16_t *) (base + 4096)),
          oq ((const SMOL_AS_LDS uint16_t *) (base + 6656)),
          ts ((const SMOL_AS_LDS uint8_t *) (base + 4608)) {}
#endif

    __device__ __forceinline__ LutLds (const uint8_t *base, LutAtLds0)
#ifdef SMOL_KERNEL_EMU
        : LutLds (base) {}
#else
        : p8 (reinterpret_cast<const SMOL_AS_LDS uint32_t *> (0u)),
          p8l (reinterpret_cast<const SMOL_AS_LDS uint32_t *> (1024u)),
          p16 (reinterpret_cast<const SMOL_AS_LDS uint32_t *> (2048u)),
          p16l (reinterpret_cast<const SMOL_AS_LDS uint32_t *> (3072u)),
          fs (reinterpret_cast<const SMOL_AS_LDS uint16_t *> (4096u)),
          oq (reinterpret_cast<const SMOL_AS_LDS uint16_t *> (6656u)),
          ts (reinterpret_cast<const SMOL_AS_LDS uint8_t *> (4608u)) { (void) base; }
#endif

    __device__ __forceinline__ uint32_t inv_p8 (uint32_t a) const { return p8 [a]; }
    __device__ __forceinline__ uint32_t inv_p8l (uint32_t a) const { return p8l [a]; }
    __device__ __forceinline__ uint32_t inv_p16 (uint32_t a) const { return p16 [a]; }
    __device__ __forceinline__ uint32_t inv_p16l (uint32_t a) const { return p16l [a]; }
    __device__ __forceinline__ uint32_t from_srgb (uint32_t v) const { return fs [v]; }
    __device__ __forceinline__ uint32_t to_srgb (uint32_t v) const { return ts [v]; }
    __device__ __forceinline__ uint32_t opaque_lin (uint32_t v) const { return oq [v]; }
    /* the same by BYTE offset into the table (index times the entry size, as smol_byte_x2 makes it in one instruction) */
    __device__ __forceinline__ uint32_t inv_p8_at (uint32_t b) const { return *reinterpret_cast<const SMOL_AS_LDS uint32_t *> (reinterpret_cast<const SMOL_AS_LDS uint8_t *> (p8) + b); }
    __device__ __forceinline__ uint32_t from_srgb_at (uint32_t b) const { return *reinterpret_cast<const SMOL_AS_LDS uint16_t *> (reinterpret_cast<const SMOL_AS_LDS uint8_t *> (fs) + b); }
    __device__ __forceinline__ uint32_t opaque_lin_at (uint32_t b) const { return *reinterpret_cast<const SMOL_AS_LDS uint16_t *> (reinterpret_cast<const SMOL_AS_LDS uint8_t *> (oq) + b); }
};

/* the entries' guard for LUT_AT_LDS_0 (folds away when the compiler knows the address, costs a scalar compare otherwise) */
__device__ __forceinline__ void lut_expect_lds_0 (const uint8_t *lds)
{
#ifndef SMOL_KERNEL_EMU
    if (reinterpret_cast<uintptr_t> (lds) & 0x3ffffu)
        __builtin_trap ();
#else
    (void) lds;
#endif
}

/* cooperative copy of the tables into LDS (caller synchronises afterwards) */
__device__ __forceinline__ void lut_fill_lds (uint8_t *base, uint32_t tid, uint32_t nthreads)
{
    uint32_t *d = reinterpret_cast<uint32_t *> (base);
    for (uint32_t i = tid; i < 256; i += nthreads)
    {
        d [i] = smol_inv_p8_lut [i];
        d [256 + i] = smol_inv_p8l_lut [i];
        d [512 + i] = smol_inv_p16_lut [i];
        d [768 + i] = smol_inv_p16l_lut [i];
    }
    uint16_t *f = reinterpret_cast<uint16_t *> (base + 4096);
    for (uint32_t i = tid; i < 256; i += nthreads)
        f [i] = smol_from_srgb_lut [i];
    const uint32_t *tsrc = reinterpret_cast<const uint32_t *> (smol_to_srgb_lut);
    uint32_t *t = reinterpret_cast<uint32_t *> (base + 4608);
    for (uint32_t i = tid; i < 512; i += nthreads)
        t [i] = tsrc [i];
    /* fused table for opaque pixels: exactly what unpack_word computes for alpha = 255 */
    uint16_t *o = reinterpret_cast<uint16_t *> (base + 6656);
    for (uint32_t i = tid; i < 256; i += nthreads)
        o [i] = (uint16_t) LutConst ().opaque_lin (i);
}

/* premultiplied -> unassociated 8 bit via the inverse table: :227-236, 246-259 */
template <typename L>
__device__ __forceinline__ uint32_t unpremul8 (uint32_t c, uint32_t a, const L &lut)
{
    return (inv_mul (c, lut.inv_p8 (a)) >> 13) & 0xffu;
}

__device__ __forceinline__ uint32_t unpremul8 (uint32_t c, uint32_t a)
{
    return unpremul8 (c, a, LutConst ());
}

/* one bilinear tap: q + floor ((p - q) * F / 256): :1317, :1704.
 * Lanes never exceed 16 bits (B.6 of SURVEY.md), so int32 is exact. */
__device__ __forceinline__ uint32_t lerp8 (uint32_t p, uint32_t q, uint32_t F, uint32_t mask)
{
    int32_t d = ((int32_t) p - (int32_t) q) * (int32_t) F;
    return (uint32_t) ((d >> 8) + (int32_t) q) & mask;
}

/* box normalisation: :1231-1261 (16-bit lanes for 64bpp, 32-bit lanes for 128bpp) */
template <bool WIDE>
__device__ __forceinline__ uint32_t box_scale (uint32_t acc, uint32_t mul)
{
    if (!WIDE)
        acc &= 0xffffu;
    uint64_t v = (uint64_t) acc * mul + (1u << 23);
    return (uint32_t) (v >> 24) & (WIDE ? 0xffffu : 0xffu);
}

/* ------------------------------------------------------------------------- *
 * Unpack one source pixel into internal channels                            *
 *   64bpp-class: smolscale-generic.c:353-464                                *
 *   128bpp-class: :470-752                                                  *
 * ------------------------------------------------------------------------- */
/* raw pixel word (bytes in memory order; only 3 valid bytes for 24bpp) ->
 * internal premultiplied channels c[0..2] = R,G,B, c[3] = alpha lane */
template <bool WIDE, typename L>
__device__ __forceinline__ void unpack_word (const JobDev &J, uint32_t w, uint32_t (&c) [4], const L &lut)
{
    uint32_t a;

    c [0] = (w >> (8 * J.pos_in [0])) & 0xffu;
    c [1] = (w >> (8 * J.pos_in [1])) & 0xffu;
    c [2] = (w >> (8 * J.pos_in [2])) & 0xffu;
    a = (J.bpp_in == 4) ? (w >> (8 * J.pos_in [3])) & 0xffu : 0xffu;

    if (WIDE && J.p16)
    {
#pragma unroll
        for (int i = 0; i < 3; i++)
        {
            uint32_t v = c [i];
            if (J.linear)
                v = lut.from_srgb (v);
            c [i] = __umul24 (v, a);
        }
        c [3] = (a << 8) | 0x80u;
        return;
    }

    if (WIDE && J.linear && J.bpp_in == 3)
    {
        /* 24-bit input is opaque: the whole chain below is one table look-up per channel */
#pragma unroll
        for (int i = 0; i < 3; i++)
            c [i] = lut.opaque_lin (c [i]);
    }
    else if (WIDE && J.linear)
    {
        if (smol_wave_all_opaque (a))
        {
            /* opaque pixels (most pixels of most images): un-premultiplying by 255 is the
             * identity, so premultiplied and unassociated input alike reduce to the fused table */
#pragma unroll
            for (int i = 0; i < 3; i++)
                c [i] = lut.opaque_lin (c [i]);
        }
        else
        {
#pragma unroll
            for (int i = 0; i < 3; i++)
            {
                uint32_t v = c [i];
                if (!J.unassoc_in)
                    v = unpremul8 (v, a, lut);
                c [i] = premul8l (lut.from_srgb (v), a);
            }
        }
    }
    else if (J.unassoc_in)
    {
#pragma unroll
        for (int i = 0; i < 3; i++)
            c [i] = premul8 (c [i], a);
    }

    c [3] = a;
}

template <bool WIDE>
__device__ __forceinline__ void unpack_word (const JobDev &J, uint32_t w, uint32_t (&c) [4])
{
    unpack_word<WIDE> (J, w, c, LutConst ());
}

template <bool WIDE, typename L>
__device__ __forceinline__ void fetch_px (const JobDev &J, const uint8_t *rowp, uint32_t x,
                                          uint32_t (&c) [4], const L &lut)
{
    uint32_t w;

    if (J.bpp_in == 4)
    {
        const gbyte *p = gptr (rowp) + (size_t) x * 4;

        if (J.in_dword_ok)
            w = *reinterpret_cast<const SMOL_AS_GLOBAL uint32_t *> (p);
        else
            w = (uint32_t) p [0] | ((uint32_t) p [1] << 8) | ((uint32_t) p [2] << 16) | ((uint32_t) p [3] << 24);
    }
    else
    {
        const gbyte *p = gptr (rowp) + (size_t) x * 3;

        w = (uint32_t) p [0] | ((uint32_t) p [1] << 8) | ((uint32_t) p [2] << 16);
    }

    unpack_word<WIDE> (J, w, c, lut);
}

/* ------------------------------------------------------------------------- *
 * Pack internal channels into one destination pixel                         *
 *   64bpp-class: smolscale-generic.c:758-906; 128bpp-class: :912-1164       *
 * ------------------------------------------------------------------------- */
/* internal channels -> the three colour bytes and the alpha byte of the destination pixel */
template <bool WIDE, typename L>
__device__ __forceinline__ void pack_px_channels (const JobDev &J, const uint32_t (&v) [4], const L &lut,
                                                  uint32_t (&c) [3], uint32_t &a)
{
    if (WIDE && J.p16)
        a = (v [3] >> 8) & 0xffu;
    else
        a = v [3] & 0xffu;

#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        uint32_t s = v [i];

        if (!WIDE)
        {
            c [i] = J.unassoc_out ? unpremul8 (s, a, lut) : (s & 0xffu);
        }
        else if (J.p16)
        {
            if (J.linear)
                c [i] = lut.to_srgb ((inv_mul (s, lut.inv_p16l (a)) >> 19) & 0x7ffu);
            else
                c [i] = (inv_mul (s, lut.inv_p16 (a)) >> 16) & 0xffu;
        }
        else if (!J.linear)
        {
            c [i] = J.unassoc_out ? unpremul8 (s, a, lut) : (s & 0xffu);
        }
        else if (!J.quirk24 && smol_wave_all_opaque (a))
        {
            /* opaque pixel: the inverse for 255 is 1028 = 1024 + 4, so (s * 1028) >> 10 is
             * s + (s >> 8) exactly, and premultiplying the sRGB byte by 255 is the identity:
             * premultiplied, unassociated and 24-bit outputs alike are one table look-up */
            c [i] = lut.to_srgb ((s + (s >> 8)) & 0x7ffu);
        }
        else
        {
            uint32_t u = (inv_mul (s, lut.inv_p8l (a)) >> 10) & 0x7ffu;

            if (J.bpp_out == 3)
                c [i] = J.quirk24 ? lut.to_srgb (s & 0x7ffu) : lut.to_srgb (u);
            else if (J.unassoc_out)
                c [i] = lut.to_srgb (u);
            else
                c [i] = premul8 (lut.to_srgb (u), a);
        }
    }

}

/* the 32-bit destination word of a pixel (bpp_out == 4) */
template <bool WIDE, typename L>
__device__ __forceinline__ uint32_t pack_px_word (const JobDev &J, const uint32_t (&v) [4], const L &lut)
{
    uint32_t a, c [3];
    pack_px_channels<WIDE> (J, v, lut, c, a);
    return (c [0] << (8 * J.pos_out [0])) | (c [1] << (8 * J.pos_out [1]))
         | (c [2] << (8 * J.pos_out [2])) | (a << (8 * J.pos_out [3]));
}

template <bool WIDE, typename L>
__device__ __forceinline__ void store_px (const JobDev &J, uint8_t *rowp, uint32_t x,
                                          const uint32_t (&v) [4], const L &lut)
{
    uint32_t a, c [3];

    pack_px_channels<WIDE> (J, v, lut, c, a);

    if (J.bpp_out == 4)
    {
        gbyte *p = gptr (rowp) + (size_t) x * 4;
        uint32_t w = (c [0] << (8 * J.pos_out [0])) | (c [1] << (8 * J.pos_out [1]))
                   | (c [2] << (8 * J.pos_out [2])) | (a << (8 * J.pos_out [3]));

        if (J.out_dword_ok)
        {
            *reinterpret_cast<SMOL_AS_GLOBAL uint32_t *> (p) = w;
        }
        else
        {
            p [0] = (uint8_t) w;
            p [1] = (uint8_t) (w >> 8);
            p [2] = (uint8_t) (w >> 16);
            p [3] = (uint8_t) (w >> 24);
        }
    }
    else
    {
        gbyte *p = gptr (rowp) + (size_t) x * 3;

        p [J.pos_out [0]] = (uint8_t) c [0];
        p [J.pos_out [1]] = (uint8_t) c [1];
        p [J.pos_out [2]] = (uint8_t) c [2];
    }
}

/* ------------------------------------------------------------------------- *
 * 128bpp class: conversions with the flavour fixed at COMPILE time           *
 *                                                                            *
 * unpack_word<true> / pack_px_channels<true> above decide the flavour from   *
 * the job's flags at every pixel: a dozen scalar branches per pixel, each    *
 * ending the compiler's scheduling region, so that a lane converts its       *
 * pixels one after the other with an LDS round trip (table look-up, wait,    *
 * multiply, look-up, wait) between any two instructions -- 46 vector and 19  *
 * scalar instructions per source pixel in the gamma-correct 4:1 case         *
 * (profiles/r03g_sq_srgb.txt). The kernels that matter switch ONCE per lane  *
 * (wide_unpack_kind / wide_pack_kind) into code specialised by these         *
 * functors: no branches, the look-ups of all the pixels a lane holds in      *
 * flight together. Same arithmetic, same tables (:470-752, :912-1164).       *
 * ------------------------------------------------------------------------- */
enum : int { WU_ANY = 0,       /* run-time flags (unpack_word<true>): compressed 128bpp, anything unusual        */
             WU_P16,           /* premul16, no tables: c * a, A = (a << 8) | 0x80                    :616-625    */
             WU_P16L,          /* premul16 + linear: from_srgb (c) * a                               :627-660    */
             WU_LIN_P,         /* linear, premultiplied 32-bit input: unpremul8 -> from_srgb -> premul8l :481-535 */
             WU_LIN_U,         /* linear, unassociated 32-bit input: from_srgb -> premul8l           :588-614    */
             WU_LIN_O,         /* linear, every pixel opaque (24-bit input; or the caller saw alpha = 255 on the
                                * whole wave): one fused look-up per channel                                     */
             WU_FLAT_P };      /* compressed 128bpp (boxes wider than 255 pixels), premultiplied 32-bit input: the bytes
                                * as they are (:470-479); chosen by the box kernels only                           */
enum : int { WP_ANY = 0, WP_P16, WP_P16L,
             WP_LIN_P,         /* linear -> premultiplied 32-bit: unpremul8l -> to_srgb -> premul8   :1096-1109  */
             WP_LIN_U,         /* linear -> unassociated 32-bit, or 24-bit through the `321' packer  :1122-1134, :1010-1023 */
             WP_LIN_Q,         /* linear -> 24-bit through the `123' packer: to_srgb of the raw lane :922-935 (quirk24) */
             WP_LIN_O };       /* WP_LIN_P / WP_LIN_U when the caller saw alpha = 255 on the whole wave: the inverse for 255 is
                                * 1028 = 1024 + 4, (s * 1028) >> 10 = s + (s >> 8), premultiplying by 255 is the identity */

__device__ __forceinline__ int wide_unpack_kind (const JobDev &J)
{
    if (J.p16)
        return J.bpp_in == 4 ? (J.linear ? WU_P16L : WU_P16) : WU_ANY;
    if (!J.linear)
        return WU_ANY;
    if (J.bpp_in == 3)
        return WU_LIN_O;
    return J.unassoc_in ? WU_LIN_U : WU_LIN_P;
}

__device__ __forceinline__ int wide_pack_kind (const JobDev &J)
{
    if (J.p16)
        return J.bpp_out == 4 ? (J.linear ? WP_P16L : WP_P16) : WP_ANY;
    if (!J.linear)
        return WP_ANY;
    if (J.bpp_out == 3)
        return J.quirk24 ? WP_LIN_Q : WP_LIN_U;
    return J.unassoc_out ? WP_LIN_U : WP_LIN_P;
}

/* (byte B of w) << `sh' (sh in a register): the address of a table entry, in one SDWA shift */
template <int B>
__device__ __forceinline__ uint32_t smol_byte_x2 (uint32_t w, uint32_t one)
{
#ifdef SMOL_KERNEL_EMU
    return ((w >> (8 * B)) & 0xffu) << one;
#else
    uint32_t r;
    if (B == 0)
        asm ("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v" (r) : "v" (one), "v" (w));
    else if (B == 1)
        asm ("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v" (r) : "v" (one), "v" (w));
    else if (B == 2)
        asm ("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v" (r) : "v" (one), "v" (w));
    else
        asm ("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v" (r) : "v" (one), "v" (w));
    return r;
#endif
}

/* w rotated right by one byte (v_alignbit_b32) */
__device__ __forceinline__ uint32_t smol_rotr8 (uint32_t w)
{
    return (w >> 8) | (w << 24);
}

/* byte B of w */
template <int B>
__device__ __forceinline__ uint32_t smol_byte (uint32_t w)
{
    return B == 3 ? w >> 24 : (w >> (8 * B)) & 0xffu;   /* (B = 0: v_and, B = 3: v_lshrrev -- the 2-cycle class; else one v_bfe) */
}

/* (byte B of w) * m, m < 2^24: one SDWA multiply, no extraction */
template <int B>
__device__ __forceinline__ uint32_t smol_byte_mul24 (uint32_t w, uint32_t m)
{
#ifdef SMOL_KERNEL_EMU
    return ((w >> (8 * B)) & 0xffu) * (m & 0xffffffu);
#else
    uint32_t r;
    if (B == 0)
        asm ("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v" (r) : "v" (w), "v" (m));
    else if (B == 1)
        asm ("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v" (r) : "v" (w), "v" (m));
    else if (B == 2)
        asm ("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v" (r) : "v" (w), "v" (m));
    else
        asm ("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v" (r) : "v" (w), "v" (m));
    return r;
#endif
}

template <int K>
struct WideUnpack
{
    static constexpr int KIND = K;
    const JobDev &J;
    const LutLds &lut;
    uint32_t sr, sg, sb, sa;

    __device__ __forceinline__ WideUnpack (const JobDev &job, const LutLds &l)
        : J (job), lut (l), sr (8u * job.pos_in [0]), sg (8u * job.pos_in [1]), sb (8u * job.pos_in [2]),
          sa (8u * (job.pos_in [3] & 3u)) {}

    /* w: the pixel's bytes in memory order (24-bit pixels: the low three) */
    __device__ __forceinline__ void operator() (uint32_t w, uint32_t (&c) [4]) const
    {
        if (K == WU_ANY)
        {
            unpack_word<true> (J, w, c, lut);
            return;
        }
        const uint32_t r = (w >> sr) & 0xffu, g = (w >> sg) & 0xffu, b = (w >> sb) & 0xffu;
        if (K == WU_FLAT_P)
        {
            c [0] = r; c [1] = g; c [2] = b;
            c [3] = (w >> sa) & 0xffu;
            return;
        }
        if (K == WU_LIN_O)
        {
            c [0] = lut.opaque_lin (r);
            c [1] = lut.opaque_lin (g);
            c [2] = lut.opaque_lin (b);
            c [3] = 255u;
            return;
        }
        const uint32_t a = (w >> sa) & 0xffu;
        if (K == WU_P16)
        {
            c [0] = __umul24 (r, a);
            c [1] = __umul24 (g, a);
            c [2] = __umul24 (b, a);
            c [3] = (a << 8) | 0x80u;
        }
        else if (K == WU_P16L)
        {
            c [0] = __umul24 (lut.from_srgb (r), a);
            c [1] = __umul24 (lut.from_srgb (g), a);
            c [2] = __umul24 (lut.from_srgb (b), a);
            c [3] = (a << 8) | 0x80u;
        }
        else
        {
            /* premul8l (l, a) = ((l + 1) * (8a + 1) - 1) >> 11 = (l * (8a + 1) + 8a) >> 11, below 2^11 as it stands
             * (l <= 2047: at most 2048 * 2041 - 1 before the shift) */
            const uint32_t a8 = a << 3, a81 = a8 | 1u;
            uint32_t u0 = r, u1 = g, u2 = b;
            if (K == WU_LIN_P)
            {
                const uint32_t inv = lut.inv_p8 (a);
                u0 = (inv_mul (r, inv) >> 13) & 0xffu;
                u1 = (inv_mul (g, inv) >> 13) & 0xffu;
                u2 = (inv_mul (b, inv) >> 13) & 0xffu;
            }
            c [0] = (__umul24 (lut.from_srgb (u0), a81) + a8) >> 11;
            c [1] = (__umul24 (lut.from_srgb (u1), a81) + a8) >> 11;
            c [2] = (__umul24 (lut.from_srgb (u2), a81) + a8) >> 11;
            c [3] = a;
        }
    }
};

/* "every one of these pixel words is opaque, on every active lane": lets a lane take WU_LIN_O for a whole batch
 * of 32-bit pixels (photographs with an alpha channel). AND of the words, one ballot. */
__device__ __forceinline__ bool wide_batch_opaque (const JobDev &J, uint32_t and_of_words)
{
    return smol_wave_all_opaque ((and_of_words >> (8u * (J.pos_in [3] & 3u))) & 0xffu);
}

/* v_perm selector with byte `pos' of the result taken from source byte `code' (0-3: S1, 4-7: S0; 0x0c elsewhere = 0) */
__device__ __forceinline__ uint32_t wide_sel (uint32_t sel, uint32_t pos, uint32_t code)
{
    return (sel & ~(0xffu << (8u * pos))) | (code << (8u * pos));
}

template <int K>
struct WidePack
{
    const JobDev &J;
    const LutLds &lut;
    uint32_t dr, dg, db, da;
    /* the destination word is assembled by two v_perm + one OR from the registers the last arithmetic step left its
     * byte in (byte B of each): perm (c0, c1, sel01) | perm (c2, alpha, sel2a) -- no shifts by 8, 16, 24, no masks */
    uint32_t sel01, sel2a;
    static constexpr uint32_t B = (K == WP_LIN_P) ? 1u : (K == WP_P16) ? 2u : 0u;

    __device__ __forceinline__ WidePack (const JobDev &job, const LutLds &l)
        : J (job), lut (l), dr (8u * job.pos_out [0]), dg (8u * job.pos_out [1]), db (8u * job.pos_out [2]),
          da (8u * (job.pos_out [3] & 3u))
    {
        sel01 = wide_sel (wide_sel (0x0c0c0c0cu, job.pos_out [0], 4u + B), job.pos_out [1], B);
        sel2a = wide_sel (0x0c0c0c0cu, job.pos_out [2], 4u + B);
        if (job.bpp_out == 4)
            sel2a = wide_sel (sel2a, job.pos_out [3] & 3u, 0u);
    }

    __device__ __forceinline__ uint32_t assemble (uint32_t c0, uint32_t c1, uint32_t c2, uint32_t a) const
    {
        return __builtin_amdgcn_perm (c0, c1, sel01) | __builtin_amdgcn_perm (c2, a, sel2a);
    }

    /* internal channels -> the destination pixel's bytes in memory order (bpp_out == 3: the low three bytes) */
    __device__ __forceinline__ uint32_t operator() (const uint32_t (&v) [4]) const
    {
        if (K == WP_ANY)
        {
            uint32_t a, c [3];
            pack_px_channels<true> (J, v, lut, c, a);
            const uint32_t w = (c [0] << dr) | (c [1] << dg) | (c [2] << db);
            return J.bpp_out == 4 ? w | (a << da) : w;
        }
        if (K == WP_P16 || K == WP_P16L)
        {
            const uint32_t a = (v [3] >> 8) & 0xffu;
            if (K == WP_P16)
            {
                /* (s * inv) >> 16 & 0xff: byte 2 of the product */
                const uint32_t inv = lut.inv_p16 (a);
                return assemble (inv_mul (v [0], inv), inv_mul (v [1], inv), inv_mul (v [2], inv), a);
            }
            const uint32_t inv = lut.inv_p16l (a);
            return assemble (lut.to_srgb ((inv_mul (v [0], inv) >> 19) & 0x7ffu),
                             lut.to_srgb ((inv_mul (v [1], inv) >> 19) & 0x7ffu),
                             lut.to_srgb ((inv_mul (v [2], inv) >> 19) & 0x7ffu), a);
        }
        if (K == WP_LIN_O)
            return assemble (lut.to_srgb ((v [0] + (v [0] >> 8)) & 0x7ffu), lut.to_srgb ((v [1] + (v [1] >> 8)) & 0x7ffu),
                             lut.to_srgb ((v [2] + (v [2] >> 8)) & 0x7ffu), 0xffu);
        const uint32_t a = v [3] & 0xffu;
        if (K == WP_LIN_Q)
            return assemble (lut.to_srgb (v [0] & 0x7ffu), lut.to_srgb (v [1] & 0x7ffu), lut.to_srgb (v [2] & 0x7ffu), a);
        const uint32_t inv = lut.inv_p8l (a);
        uint32_t c0 = lut.to_srgb ((inv_mul (v [0], inv) >> 10) & 0x7ffu);
        uint32_t c1 = lut.to_srgb ((inv_mul (v [1], inv) >> 10) & 0x7ffu);
        uint32_t c2 = lut.to_srgb ((inv_mul (v [2], inv) >> 10) & 0x7ffu);
        if (K == WP_LIN_P)
        {
            /* premul8 (s, a) = ((s + 1) * (a + 1) - 1) >> 8 = (s * (a + 1) + a) >> 8, at most 65535 before the shift:
             * byte 1 of the sum */
            const uint32_t a1 = a + 1u;
            c0 = __umul24 (c0, a1) + a;
            c1 = __umul24 (c1, a1) + a;
            c2 = __umul24 (c2, a1) + a;
        }
        return assemble (c0, c1, c2, a);
    }
};

/* the destination word of WidePack at pixel x of the row at rowp */
__device__ __forceinline__ void wide_store_word (const JobDev &J, gbyte *rowp, uint32_t x, uint32_t w)
{
    if (J.bpp_out == 4)
    {
        gbyte *p = rowp + (size_t) x * 4;
        if (J.out_dword_ok)
            *reinterpret_cast<SMOL_AS_GLOBAL uint32_t *> (p) = w;
        else
        {
            p [0] = (uint8_t) w; p [1] = (uint8_t) (w >> 8); p [2] = (uint8_t) (w >> 16); p [3] = (uint8_t) (w >> 24);
        }
    }
    else
    {
        gbyte *p = rowp + (size_t) x * 3;
        p [0] = (uint8_t) w; p [1] = (uint8_t) (w >> 8); p [2] = (uint8_t) (w >> 16);
    }
}

/* ------------------------------------------------------------------------- *
 * Closed-form sample positions (host twins: smol_plan.c)                    *
 * ------------------------------------------------------------------------- */
__device__ __forceinline__ void bilin_tap (uint64_t pos, uint32_t dim_in, uint32_t &ofs, uint32_t &F)
{
    uint32_t o = (uint32_t) (pos >> 32);

    if (o >= dim_in - 1)
    {
        ofs = dim_in - 2;
        F = 0;
    }
    else
    {
        ofs = o;
        F = 256u - ((uint32_t) (pos >> 24) & 0xffu);
    }
}

__device__ __forceinline__ void box_span (uint64_t step, uint32_t dim_in, uint32_t k,
                                          uint32_t &o0, uint32_t &o1, uint32_t &f_right, uint32_t &w_left)
{
    uint64_t b = (uint64_t) k * step;
    uint64_t e = b + step;
    uint32_t last = (uint32_t) (e >> 16);

    o0 = (uint32_t) (b >> 16);
    o1 = last > dim_in - 1 ? dim_in - 1 : last;
    f_right = (uint32_t) (e >> 8) & 0xffu;
    w_left = k ? 255u - ((uint32_t) (b >> 8) & 0xffu) : 256u;
}

/* ------------------------------------------------------------------------- *
 * Horizontal result H(row, x): the value the reference's hfilter leaves in   *
 * its scaled row (smolscale-generic.c:1290-1611), computed on demand.       *
 * A single loop over "source taps" keeps one fetch_px call site.            *
 * ------------------------------------------------------------------------- */
template <bool WIDE, typename L>
__device__ void hsample (const JobDev &J, uint32_t row, uint32_t x, uint32_t (&out) [4], const L &lut)
{
    const uint8_t *rowp = J.src + (size_t) row * J.src_stride;
    const uint32_t mask = WIDE ? 0x00ffffffu : 0xffu;
    uint32_t acc [4] = { 0, 0, 0, 0 };
    uint32_t px [4], prev [4] = { 0, 0, 0, 0 };

    if (J.filter_h == F_BILINEAR)
    {
        const uint32_t h = J.halv_h;
        const uint32_t ntaps = 2u << h;
        uint64_t pos = J.frac0_x + (uint64_t) (x << h) * J.step_x;
        uint32_t ofs = 0, F = 0;

        for (uint32_t t = 0; t < ntaps; t++)
        {
            if (!(t & 1))
            {
                bilin_tap (pos, J.w_in, ofs, F);
                pos += J.step_x;
            }

            fetch_px<WIDE> (J, rowp, ofs + (t & 1), px, lut);

            if (t & 1)
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc [i] += lerp8 (prev [i], px [i], F, mask);
            }
            else
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    prev [i] = px [i];
            }
        }

#pragma unroll
        for (int i = 0; i < 4; i++)
            out [i] = (acc [i] >> h) & mask;
    }
    else if (J.filter_h == F_BOX)
    {
        uint32_t o0, o1, fr, wl;

        box_span (J.step_x, J.w_in, x, o0, o1, fr, wl);

        for (uint32_t s = o0; s <= o1; s++)
        {
            fetch_px<WIDE> (J, rowp, s, px, lut);

            if (s == o0)
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc [i] += ((px [i] * wl) >> 8) & mask;
            }
            else if (s == o1)
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc [i] += ((px [i] * fr) >> 8) & mask;
            }
            else
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc [i] += px [i];
            }
        }

#pragma unroll
        for (int i = 0; i < 4; i++)
            out [i] = box_scale<WIDE> (acc [i], J.span_mul_x);
    }
    else
    {
        fetch_px<WIDE> (J, rowp, J.filter_h == F_ONE ? 0u : x, out, lut);
    }
}

/* ------------------------------------------------------------------------- *
 * k_general: one thread = one output pixel                                  *
 *   vertical bilinear: smolscale-generic.c:1648-2007                        *
 *   vertical box 64bpp: :2026-2161, 128bpp: :2163-2260                      *
 *   copy / one: :2262-2318                                                  *
 * ------------------------------------------------------------------------- */
template <bool WIDE, typename L>
__device__ void general_pixel (const JobDev &J, uint32_t x, uint32_t y, const L &lut)
{
    const uint32_t mask = WIDE ? 0x00ffffffu : 0xffu;
    uint32_t acc [4] = { 0, 0, 0, 0 };
    uint32_t v [4], top [4] = { 0, 0, 0, 0 };

    if (J.filter_v == F_BILINEAR)
    {
        const uint32_t h = J.halv_v;
        const uint32_t ntaps = 2u << h;
        uint64_t pos = J.frac0_y + (uint64_t) (y << h) * J.step_y;
        uint32_t oy = 0, F = 0;

        for (uint32_t t = 0; t < ntaps; t++)
        {
            if (!(t & 1))
            {
                bilin_tap (pos, J.h_in, oy, F);
                pos += J.step_y;
            }

            hsample<WIDE> (J, oy + (t & 1), x, v, lut);

            if (t & 1)
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc [i] += lerp8 (top [i], v [i], F, mask);
            }
            else
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    top [i] = v [i];
            }
        }

#pragma unroll
        for (int i = 0; i < 4; i++)
            v [i] = (acc [i] >> h) & mask;
    }
    else if (J.filter_v == F_BOX)
    {
        uint32_t y0, y1, w2, w1;

        box_span (J.step_y, J.h_in, y, y0, y1, w2, w1);

        /* 128bpp-class rows weight the last row by w2 - 1 and skip it when
         * w2 == 0 (:2236-2253); 64bpp-class rows use w2 as is (:2042-2069). */
        uint32_t last_w = WIDE ? (w2 ? w2 - 1 : 0) : w2;
        uint32_t y_end = (WIDE && w2 == 0) ? y1 - 1 : y1;

        for (uint32_t r = y0; r <= y_end; r++)
        {
            hsample<WIDE> (J, r, x, v, lut);

            if (r == y0)
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc [i] += ((v [i] * w1) >> 8) & mask;
            }
            else if (r == y1)
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc [i] += ((v [i] * last_w) >> 8) & mask;
            }
            else
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc [i] += v [i];
            }
        }

#pragma unroll
        for (int i = 0; i < 4; i++)
            v [i] = box_scale<WIDE> (acc [i], J.span_mul_y);
    }
    else
    {
        hsample<WIDE> (J, J.filter_v == F_ONE ? 0u : y, x, v, lut);
    }

    store_px<WIDE> (J, J.dst + (size_t) (y - J.first_row) * J.dst_stride, x, v, lut);
}

constexpr int GENERAL_BLOCK = 256;

template <bool WIDE>
__device__ __forceinline__ void general_body (const JobDev &J)
{
    const uint32_t tiles_x = (J.w_out + GENERAL_BLOCK - 1) / GENERAL_BLOCK;
    const uint32_t row = blockIdx.x / tiles_x;
    const uint32_t x = (blockIdx.x % tiles_x) * GENERAL_BLOCK + threadIdx.x;

    if (WIDE)
    {
        /* the 128bpp class looks tables up several times per pixel: workgroup copy in LDS */
        __shared__ __attribute__ ((aligned (16))) uint8_t lut_lds [LUT_LDS_BYTES];
        if (row >= J.n_rows)
            return;                                 /* uniform for the whole workgroup */
#ifdef SMOL_KERNEL_EMU
        lut_fill_lds (lut_lds, 0, 1);               /* emulated lanes run one after the other */
#else
        lut_fill_lds (lut_lds, threadIdx.x, GENERAL_BLOCK);
        __syncthreads ();
#endif
        if (x < J.w_out)
            general_pixel<WIDE> (J, x, J.first_row + row, LutLds (lut_lds));
    }
    else
    {
        if (row >= J.n_rows || x >= J.w_out)
            return;
        general_pixel<WIDE> (J, x, J.first_row + row, LutConst ());
    }
}

/* batch form: one descriptor per blockIdx.y, descriptors in device memory */
template <bool WIDE>
__global__ void __launch_bounds__ (GENERAL_BLOCK)
k_general_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.y);
    general_body<WIDE> (J);
}

/* single-job form: descriptor passed by value in the kernel arguments */
template <bool WIDE>
__global__ void __launch_bounds__ (GENERAL_BLOCK)
k_general_one (const JobDev J)
{
    general_body<WIDE> (J);
}

/* ------------------------------------------------------------------------- *
 * k_wavepx: one WAVE per output pixel, for vertical box filters whose output  *
 * is too small to fill the chip with one lane per pixel (e.g. 8192^2 -> 30^2: *
 * 900 pixels, each the sum of ~273 x 273 sources). The 64 lanes take the rows *
 * of the pixel's vertical span in turn (each row's horizontal result is the   *
 * reference's truncated per-row value, so the row sums are exact integers and *
 * their order does not matter) and meet through a shuffle butterfly.          *
 * ------------------------------------------------------------------------- */
#ifndef SMOL_KERNEL_EMU
__device__ __forceinline__ uint32_t wave_sum (uint32_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v += (uint32_t) __shfl_xor ((int) v, m, 64);
    return v;
}
#endif

template <bool WIDE>
__device__ __forceinline__ void wavepx_body (const JobDev &J)
{
    const uint32_t mask = WIDE ? 0x00ffffffu : 0xffu;
    const uint32_t lane = threadIdx.x;
    const uint32_t idx = blockIdx.x;
    const uint32_t row = idx / J.w_out;
    const uint32_t x = idx - row * J.w_out;

    __shared__ __attribute__ ((aligned (16))) uint8_t lut_lds [LUT_LDS_BYTES];
    if (row >= J.n_rows)
        return;
    lut_fill_lds (lut_lds, lane, 64);
    __syncthreads ();
    const LutLds lut (lut_lds);

    const uint32_t y = J.first_row + row;
    uint32_t y0, y1, w2, w1;
    box_span (J.step_y, J.h_in, y, y0, y1, w2, w1);
    const uint32_t last_w = WIDE ? (w2 ? w2 - 1 : 0) : w2;          /* see general_pixel */
    const uint32_t y_end = (WIDE && w2 == 0) ? y1 - 1 : y1;

    uint32_t acc [4] = { 0, 0, 0, 0 };
    uint32_t v [4];

    for (uint32_t r = y0 + lane; r <= y_end; r += 64)
    {
        hsample<WIDE> (J, r, x, v, lut);
        const uint32_t w = (r == y0) ? w1 : ((r == y1) ? last_w : 0x10000u);
#pragma unroll
        for (int i = 0; i < 4; i++)
            acc [i] += (w == 0x10000u) ? v [i] : (((v [i] * w) >> 8) & mask);
    }

#pragma unroll
    for (int i = 0; i < 4; i++)
        acc [i] = wave_sum (acc [i]);

    if (lane == 0)
    {
#pragma unroll
        for (int i = 0; i < 4; i++)
            v [i] = box_scale<WIDE> (acc [i], J.span_mul_y);
        store_px<WIDE> (J, J.dst + (size_t) row * J.dst_stride, x, v, lut);
    }
}

template <bool WIDE>
__global__ void __launch_bounds__ (64)
k_wavepx_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.y);
    wavepx_body<WIDE> (J);
}

template <bool WIDE>
__global__ void __launch_bounds__ (64)
k_wavepx_one (const JobDev J)
{
    wavepx_body<WIDE> (J);
}

/* ------------------------------------------------------------------------- *
 * k_pow2: exact 2^n : 1 bilinear-with-halvings, 32-bit premultiplied pixels *
 *                                                                           *
 * With w_in == w_out << (HX+1) every horizontal sample j sits at offset 2j  *
 * with weight F = 128 (closed form of smolscale-generic.c:24-52), so a tap  *
 *      q + (((p - q) * 128) >> 8)  ==  floor ((p + q) / 2)   per channel,   *
 * and the halving step is (sum of 2^HX taps) >> HX (:1363-1387). The same   *
 * holds vertically on the 8-bit horizontal results (:1684-1811). All four   *
 * channels of a pixel stay packed in one 32-bit register.                   *
 * ------------------------------------------------------------------------- */

/* per-byte floor average of two packed pixels */
__device__ __forceinline__ uint32_t avg4 (uint32_t a, uint32_t b)
{
    return (a & b) + (((a ^ b) >> 1) & 0x7f7f7f7fu);
}

/* (a + b + c + d) >> 2 per byte, exact (16-bit lanes) */
__device__ __forceinline__ uint32_t sum4_shift2 (uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const uint32_t m = 0x00ff00ffu;
    uint32_t lo = (a & m) + (b & m) + (c & m) + (d & m);
    uint32_t hi = ((a >> 8) & m) + ((b >> 8) & m) + ((c >> 8) & m) + ((d >> 8) & m);
    return ((lo >> 2) & m) | (((hi >> 2) & m) << 8);
}

/* Reduce 2^(H+1) consecutive packed pixels px[0..] to one: taps then halving. */
template <int H>
__device__ __forceinline__ uint32_t reduce_pow2 (const uint32_t *px)
{
    if (H == 0)
        return avg4 (px [0], px [1]);
    if (H == 1)
        return avg4 (avg4 (px [0], px [1]), avg4 (px [2], px [3]));
    return sum4_shift2 (avg4 (px [0], px [1]), avg4 (px [2], px [3]),
                        avg4 (px [4], px [5]), avg4 (px [6], px [7]));
}

template <int HX, int HY>
struct Pow2Cfg
{
    static constexpr int RX = 2 << HX;                 /* source px per output px, x */
    static constexpr int RY = 2 << HY;                 /* source rows per output row */
    static constexpr int OPT = RX >= 4 ? 1 : 4 / RX;   /* output px per thread       */
    static constexpr int NLOAD = OPT * RX / 4;         /* 16-byte loads per src row  */
    static constexpr int BLOCK_X = 64;                 /* one wave = 64 thread-columns of one output row */
    static constexpr int BLOCK_Y = 16;                 /* output rows per workgroup                      */
};

typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
typedef u32x4 u32x4a4 __attribute__ ((aligned (4)));       /* a 16-byte load unit on a 4-byte boundary (cropped images) */
typedef uint32_t u32a1 __attribute__ ((aligned (1)));      /* a dword at any byte address (global memory needs no alignment on gfx950) */
typedef uint32_t u32x2a1 __attribute__ ((ext_vector_type (2), aligned (1)));

/* value of `v' in the NEXT lane of this lane's quad (lane 3 of a quad: its own): one DPP move */
__device__ __forceinline__ uint32_t smol_quad_next (uint32_t v)
{
#ifdef SMOL_KERNEL_EMU
    return smol_shfl_idx (v, (threadIdx.x & 3u) == 3u ? threadIdx.x : threadIdx.x + 1u);
#else
    return (uint32_t) __builtin_amdgcn_mov_dpp ((int) v, 0xF9 /* quad_perm:[1,2,3,3] */, 0xf, 0xf, true);
#endif
}

typedef uint32_t u32x2 __attribute__ ((ext_vector_type (2)));
typedef u32x2 u32x2a4 __attribute__ ((aligned (4)));       /* an 8-byte store unit on a 4-byte boundary */

/* 16-byte NON-TEMPORAL output store on a dword-aligned address (k_up, bound by its stores: aligned rows 92 us
 * against 120 us with plain stores on config 3, rows that start anywhere 106 against 111; profiles/r04_store_policy_ab.txt).
 * Never put this and a plain store on the two sides of a branch: the compiler merges them into ONE plain store (the hint
 * is metadata, and metadata that differs is dropped). */
__device__ __forceinline__ void store_nt128 (gbyte *p, u32x4 v)
{
#if defined (SMOL_KERNEL_EMU)
    __builtin_memcpy (p, &v, 16);
#else
    __builtin_nontemporal_store (v, reinterpret_cast<SMOL_AS_GLOBAL u32x4a4 *> (p));
#endif
}

/* 16-byte write-through output store (see store_out32) */
__device__ __forceinline__ void store_out128 (gbyte *p, u32x4 v)
{
#if defined (SMOL_KERNEL_EMU)
    __builtin_memcpy (p, &v, 16);       /* (the address is only dword aligned: smol_jobdev.h out_vec16_ok) */
#elif defined (SMOL_STORES_NT)
    __builtin_nontemporal_store (v, reinterpret_cast<SMOL_AS_GLOBAL u32x4a4 *> (p));
#elif defined (SMOL_STORES_PLAIN)
    *reinterpret_cast<SMOL_AS_GLOBAL u32x4a4 *> (p) = v;
#else
    asm volatile ("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v" (p), "v" (v) : "memory");
#endif
}

/* the same, non-temporal (the nt bit): for runs of 16-byte stores on rows that start anywhere in a line (k_up) */
__device__ __forceinline__ void store_out128_nt (gbyte *p, u32x4 v)
{
#if defined (SMOL_KERNEL_EMU)
    __builtin_memcpy (p, &v, 16);
#else
    asm volatile ("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v" (p), "v" (v) : "memory");
#endif
}

/* twelve bytes at any address as an 8-byte and a 4-byte NON-TEMPORAL store (k_up's 24-bit rows). The builtin, not an asm
 * statement: the compiler then addresses the row as scalar base + 32-bit lane offset, which the asm forms (a 64-bit
 * address per lane) lose 5-10 % to (profiles/r05_up_rgb8.txt). Never beside a plain store on the other side of a branch
 * (see store_nt128). */
__device__ __forceinline__ void store_out96_nt (gbyte *p, uint32_t d0, uint32_t d1, uint32_t d2)
{
#if defined (SMOL_KERNEL_EMU)
    const uint32_t d [3] = { d0, d1, d2 };
    __builtin_memcpy (p, d, 12);
#else
    __builtin_nontemporal_store (u32x2a1 { d0, d1 }, reinterpret_cast<SMOL_AS_GLOBAL u32x2a1 *> (p));
    __builtin_nontemporal_store (d2, reinterpret_cast<SMOL_AS_GLOBAL u32a1 *> (p + 8));
#endif
}

/* one dword at row + off (any byte offset), non-temporal. Only where the choice is a compile-time one (k_lind's NT
 * flavour): beside a plain store on the other side of a run-time branch the compiler merges the two into one plain store. */
__device__ __forceinline__ void store_nt32_at (gbyte *row, uint32_t off, uint32_t w)
{
#if defined (SMOL_KERNEL_EMU)
    __builtin_memcpy (row + off, &w, 4);
#else
    __builtin_nontemporal_store (w, reinterpret_cast<SMOL_AS_GLOBAL u32a1 *> (row + off));
#endif
}

/* workgroup -> (x, y, job) under the XCD-aware order (smol_jobdev.h: TileSwz) */
__device__ __forceinline__ void tile_coords (const TileSwz &S, uint32_t &bx, uint32_t &by, uint32_t &bz)
{
    if (!S.on)
    {
        bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
        return;
    }
    const uint32_t o = blockIdx.x, xcd = o & 7u, slot = o >> 3;
    const uint32_t w = (xcd < S.r ? xcd * (S.q + 1u) : S.r * (S.q + 1u) + (xcd - S.r) * S.q) + slot;
    bz = S.inv_gxgy ? (uint32_t) (((uint64_t) w * S.inv_gxgy) >> 32) : w;      /* 0: divide by one */
    const uint32_t rem = w - bz * S.gxgy;
    by = S.inv_gx ? (uint32_t) (((uint64_t) rem * S.inv_gx) >> 32) : rem;
    bx = rem - by * S.gx;
}

template <int HX, int HY>
__device__ __forceinline__ void pow2_body (const JobDev &J, const uint32_t bx, const uint32_t by)
{
    using C = Pow2Cfg<HX, HY>;

    /* Workgroup = 64 thread-columns x 16 output rows, one wave per output row:
     * measured best of the layouts tried (tools/microbench/pow2_variants.hip, 83.9 % of
     * 8 TB/s against 79.5 % for 256 x 1), with non-temporal loads AND stores. */
    /* grid = (column tiles, row tiles, jobs): no division at the head of a workgroup */
    const uint32_t tcols = J.w_out / C::OPT;
    const uint32_t row = by * blockDim.y + threadIdx.y;
    const uint32_t tc = bx * C::BLOCK_X + threadIdx.x;

    if (row >= J.n_rows || tc >= tcols)
        return;

    const uint32_t y = J.first_row + row;
    const gbyte *sp = gptr (J.src) + (size_t) y * C::RY * J.src_stride + (size_t) tc * (C::NLOAD * 16);

    /* Issue every load of the block first: RY * NLOAD independent 16-byte loads
     * in flight per lane. */
    u32x4 raw [C::RY][C::NLOAD];
#pragma unroll
    for (int r = 0; r < C::RY; r++)
#pragma unroll
        for (int l = 0; l < C::NLOAD; l++)
            raw [r][l] = __builtin_nontemporal_load (
                reinterpret_cast<const SMOL_AS_GLOBAL u32x4a4 *> (sp + (size_t) r * J.src_stride) + l);

    uint32_t hres [C::OPT][C::RY];
#pragma unroll
    for (int r = 0; r < C::RY; r++)
    {
        uint32_t px [C::NLOAD * 4];
#pragma unroll
        for (int l = 0; l < C::NLOAD; l++)
        {
            px [l * 4 + 0] = raw [r][l].x;
            px [l * 4 + 1] = raw [r][l].y;
            px [l * 4 + 2] = raw [r][l].z;
            px [l * 4 + 3] = raw [r][l].w;
        }
#pragma unroll
        for (int o = 0; o < C::OPT; o++)
            hres [o][r] = reduce_pow2<HX> (&px [o * C::RX]);
    }

    uint32_t outv [C::OPT];
#pragma unroll
    for (int o = 0; o < C::OPT; o++)
    {
        uint32_t v = reduce_pow2<HY> (hres [o]);
        outv [o] = __builtin_amdgcn_perm (v, v, J.out_perm);
    }

    gbyte *dp = gptr (J.dst) + (size_t) row * J.dst_stride + (size_t) tc * (C::OPT * 4);
    if (C::OPT == 1)
        store_out32 (reinterpret_cast<SMOL_AS_GLOBAL uint32_t *> (dp), outv [0]);
    else
        store_out64 (reinterpret_cast<SMOL_AS_GLOBAL uint64_t *> (dp), outv [0], outv [C::OPT - 1]);
}

template <int HX, int HY>
__global__ void __launch_bounds__ (1024)
k_pow2_batch (const JobDev *jobs, const TileSwz swz)
{
    uint32_t bx, by, bz;
    tile_coords (swz, bx, by, bz);
    pow2_body<HX, HY> (load_job (jobs, bz), bx, by);
}

template <int HX, int HY>
__global__ void __launch_bounds__ (1024)
k_pow2_one (const JobDev J, const TileSwz swz)
{
    uint32_t bx, by, bz;
    tile_coords (swz, bx, by, bz);
    pow2_body<HX, HY> (J, bx, by);
}


} /* namespace smolhip */
