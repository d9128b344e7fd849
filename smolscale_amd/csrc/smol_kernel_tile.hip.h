/* smol_kernel_tile.hip.h -- k_tile: wave-autonomous tiled kernel for the
 * 64bpp-class bilinear family (BILINEAR 0-2H / COPY / ONE on either axis, every
 * pixel-type pair). Included at the end of smol_kernels.hip.h.
 *
 * One 64-lane workgroup (= one wave, so __syncthreads() is free) owns 64 output
 * columns x tile_rows output rows; lane = output column.
 *
 *   stage  the source rows a tile needs come into LDS a group at a time through
 *          16-byte ALIGNED global loads, contiguous across the wave, whatever the
 *          pixel size or row alignment (24bpp rows with odd strides included):
 *          each source byte is fetched once per tile. 32-bit unassociated pixels
 *          on 4-byte boundaries are premultiplied right here, once per pixel
 *          (smolscale-generic.c:238-244, 415-464);
 *   H      each lane evaluates its column's horizontal taps from LDS
 *          (:1290-1425, :1558-1611). Values of this class are 8 bit, so a tap
 *              q + (((p - q) * F) >> 8)  ==  (p * F + q * (256 - F)) >> 8
 *          never goes negative and two channels share one register as 16-bit lanes;
 *   V      source rows arrive in ascending order, so the vertical filter
 *          (:1648-2007, :2262-2318) runs in registers per column: the lane keeps
 *          the previous row's H value and emits an output pixel as soon as the
 *          rows of its last sample have passed.
 *
 * The body is a template over (how a pixel is read from LDS, how many horizontal
 * samples, how a pixel is stored) so that the per-pixel code has no mode
 * branches; the kernel entry switches once, wave-uniformly, on J.tile_variant. */

#pragma once

namespace smolhip {

constexpr uint32_t TILE_COLS = 64;

/* build-time A/B switches (tools/build_tuning.sh "-DSMOL_TILE_...=0"): the batched box interior, its per-lane
 * rotated read order and the raised issue priority around the staging requests. Three experiments of round 3 that
 * measured SLOWER are not in the source any more (git 6a0830b, b1ae108, cd7fed3; numbers in DESIGN.md 4.2):
 * register-prefetch staging, software-pipelined tap reads, byte-addressed LDS reads. */
#ifndef SMOL_TILE_BOXBATCH
#define SMOL_TILE_BOXBATCH 1
#endif
#ifndef SMOL_TILE_PRIO
#define SMOL_TILE_PRIO 1        /* s_setprio around the staging requests */
#endif
#ifndef SMOL_TILE_BOXROT
#define SMOL_TILE_BOXROT 1      /* per-lane rotated read order inside a batch of four box pixels */
#endif

constexpr uint32_t PAIR_MASK = 0x00ff00ffu;

struct Px2 { uint32_t lo, hi; };     /* bytes 0,2 and bytes 1,3 of a pixel, 16-bit lanes */

__device__ __forceinline__ Px2 px2_from_word (uint32_t w)
{
    Px2 r;
    r.lo = w & PAIR_MASK;
    r.hi = __builtin_amdgcn_perm (w, w, 0x0c030c01u);     /* (w >> 8) & PAIR_MASK in one v_perm */
    return r;
}

/* Two 16-bit lanes per register; the compiler maps these to v_pk_*_u16. */
typedef unsigned short u16x2 __attribute__ ((ext_vector_type (2)));

__device__ __forceinline__ u16x2 as_u16x2 (uint32_t v) { return __builtin_bit_cast (u16x2, v); }
__device__ __forceinline__ uint32_t as_u32 (u16x2 v) { return __builtin_bit_cast (uint32_t, v); }

/* One bilinear tap for the sRGB-linear class -- 11-bit values, two per register -- in packed 16-bit arithmetic:
 *      q + (((p - q) * F) >> 8)                                        (smolscale-generic.c:1317, :1704)
 * needs 20 bits for the product, but with F = 16 * Fh + Fl and d = p - q
 *      floor (d * F / 256) = floor ((d * Fh + floor (d * Fl / 16)) / 16)
 * every intermediate fits a signed 16-bit lane: |d * Fl| <= 2047 * 15, |d * Fh + ...| <= 2047 * 15 + 1920 for
 * F <= 255 and exactly 16 * d for F = 256. Six packed instructions for two channels, no unpacking of the halves
 * (the 32-bit form: eleven). Exhaustively checked on the host over every (d, F): tests/test_emu_kernels.py. */
typedef short i16x2 __attribute__ ((ext_vector_type (2)));
__device__ __forceinline__ i16x2 as_i16x2 (uint32_t v) { return __builtin_bit_cast (i16x2, v); }
__device__ __forceinline__ uint32_t as_u32 (i16x2 v) { return __builtin_bit_cast (uint32_t, v); }

struct Lerp11W { uint32_t fh2, fl2; };          /* (F >> 4, F & 15), each in both halves */

__device__ __forceinline__ Lerp11W lerp11_weights (uint32_t F)
{
    Lerp11W w;
    w.fh2 = (F >> 4) * 0x00010001u;
    w.fl2 = (F & 15u) * 0x00010001u;
    return w;
}

__device__ __forceinline__ uint32_t lerp11_pk (uint32_t p, uint32_t q, const Lerp11W &w)
{
    const i16x2 d = as_i16x2 (p) - as_i16x2 (q);
    const i16x2 y = (d * as_i16x2 (w.fl2)) >> 4;
    const i16x2 x = (d * as_i16x2 (w.fh2) + y) >> 4;
    return as_u32 (as_i16x2 (q) + x);
}

/* (p*F + q*(256-F)) >> 8 per 16-bit lane. For 8-bit p, q and F in 0..256 the sum
 * is at most 255*256 < 2^16, so 16-bit lane arithmetic is exact and the result
 * needs no mask. */
__device__ __forceinline__ Px2 tap2 (Px2 p, Px2 q, uint32_t F)
{
    const unsigned short f = (unsigned short) F, g = (unsigned short) (256u - F);
    const u16x2 F2 = { f, f }, G2 = { g, g };
    Px2 r;
    r.lo = as_u32 ((as_u16x2 (p.lo) * F2 + as_u16x2 (q.lo) * G2) >> 8);
    r.hi = as_u32 ((as_u16x2 (p.hi) * F2 + as_u16x2 (q.hi) * G2) >> 8);
    return r;
}

/* lane-wise a + b on both 16-bit halves. Every sum this file forms stays below 2^16 per lane (at most 4 taps of
 * 8 bits; box sums of at most 256 pixels of 8 bits, smolscale-generic.c:1427-1479 relies on the same bound), so no
 * carry ever crosses the lanes and a plain 32-bit add IS the packed add -- and v_add_u32 issues in half the time of
 * v_pk_add_u16 (2 against 4 cycles per wave: tools/microbench/valu_rate.hip, profiles/r03a_valu_rate.txt). Spelled as
 * an asm statement because the compiler turns (a + b) >> 1 on packed lanes into a per-lane average otherwise. */
__device__ __forceinline__ uint32_t pk_add_u16 (uint32_t a, uint32_t b)
{
#ifdef SMOL_KERNEL_EMU
    return as_u32 (as_u16x2 (a) + as_u16x2 (b));
#elif defined (SMOL_TILE_PK_ADD)
    uint32_t r;
    asm ("v_pk_add_u16 %0, %1, %2" : "=v" (r) : "v" (a), "v" (b));
    return r;
#else
    uint32_t r;
    asm ("v_add_u32 %0, %1, %2" : "=v" (r) : "v" (a), "v" (b));
    return r;
#endif
}

__device__ __forceinline__ Px2 px2_add (Px2 a, Px2 b)
{
    Px2 r;
    r.lo = pk_add_u16 (a.lo, b.lo);
    r.hi = pk_add_u16 (a.hi, b.hi);
    return r;
}

__device__ __forceinline__ Px2 px2_shr (Px2 a, uint32_t n)
{
    const unsigned short s = (unsigned short) n;
    const u16x2 S2 = { s, s };
    Px2 r;
    r.lo = as_u32 (as_u16x2 (a.lo) >> S2);
    r.hi = as_u32 (as_u16x2 (a.hi) >> S2);
    return r;
}

/* (v * w) >> 8 per 16-bit lane, w in 0..256: edge weights of the box filter
 * (weight_pixel_64bpp smolscale-generic.c:1177-1182, weight_edge_row_64bpp :2026-2040) */
__device__ __forceinline__ Px2 px2_mulw (Px2 v, uint32_t w)
{
    const unsigned short ws = (unsigned short) w;
    const u16x2 W2 = { ws, ws };
    Px2 r;
    r.lo = as_u32 ((as_u16x2 (v.lo) * W2) >> 8);
    r.hi = as_u32 ((as_u16x2 (v.hi) * W2) >> 8);
    return r;
}

/* box normalisation of the four 16-bit sums: scale_64bpp smolscale-generic.c:1231-1245. Only bits 24-31 of
 * c * mul + 2^23 are kept, c is a 16-bit sum and mul = span_mul < 2^21 (a box axis has ratio > 8:
 * smol_plan.c plan_axis), so the low 32 bits of a 24 x 24 bit multiply are the whole answer. */
__device__ __forceinline__ uint32_t box_scale16 (uint32_t c, uint32_t mul)
{
    return ((__umul24 (c, mul) + (1u << 23)) >> 24) & 0xffu;
}

/* sum of the four byte products a.b[i] * w.b[i], plus acc (v_dot4_u32_u8) */
__device__ __forceinline__ uint32_t smol_udot4 (uint32_t a, uint32_t w, uint32_t acc)
{
#ifdef SMOL_KERNEL_EMU
    for (int i = 0; i < 4; i++)
        acc += ((a >> (8 * i)) & 0xffu) * ((w >> (8 * i)) & 0xffu);
    return acc;
#else
    return __builtin_amdgcn_udot4 (a, w, acc, false);
#endif
}

__device__ __forceinline__ Px2 px2_box_scale (Px2 a, uint32_t mul)
{
    Px2 r;
    r.lo = box_scale16 (a.lo & 0xffffu, mul) | (box_scale16 (a.lo >> 16, mul) << 16);
    r.hi = box_scale16 (a.hi & 0xffffu, mul) | (box_scale16 (a.hi >> 16, mul) << 16);
    return r;
}

/* (lo.h * w.lo + hi.h * w.hi) for the 16-bit halves h picked by `sel' (PAIR_LO: the low halves of
 * `lo' and `hi', PAIR_HI: the high ones), w = packed 16-bit weights: v_perm_b32 + v_dot2_u32_u16.
 * Exact: halves < 2^16, weights <= 256, the sum of two products fits 32 bits. */
constexpr uint32_t PAIR_LO = 0x05040100u, PAIR_HI = 0x07060302u;

__device__ __forceinline__ uint32_t pair_dot (uint32_t hi, uint32_t lo, uint32_t sel, uint32_t w)
{
    const uint32_t pr = __builtin_amdgcn_perm (hi, lo, sel);
#ifdef SMOL_KERNEL_EMU
    return (pr & 0xffffu) * (w & 0xffffu) + (pr >> 16) * (w >> 16);
#else
    typedef unsigned short us2 __attribute__ ((ext_vector_type (2)));
    return __builtin_amdgcn_udot2 (__builtin_bit_cast (us2, pr), __builtin_bit_cast (us2, w), 0u, false);
#endif
}

/* the same with the pair already side by side in one register (lo | hi << 16) */
__device__ __forceinline__ uint32_t pair_word_dot (uint32_t pr, uint32_t w)
{
#ifdef SMOL_KERNEL_EMU
    return (pr & 0xffffu) * (w & 0xffffu) + (pr >> 16) * (w >> 16);
#else
    typedef unsigned short us2 __attribute__ ((ext_vector_type (2)));
    return __builtin_amdgcn_udot2 (__builtin_bit_cast (us2, pr), __builtin_bit_cast (us2, w), 0u, false);
#endif
}

/* unassociated -> premultiplied on a whole pixel word; ain = alpha byte index */
__device__ __forceinline__ uint32_t premul_word (uint32_t w, uint32_t ain)
{
    const uint32_t a = (w >> (8 * ain)) & 0xffu;
    const uint32_t m = a + 1u;
    uint32_t lo = w & PAIR_MASK;
    uint32_t hi = (w >> 8) & PAIR_MASK;

    /* full 32-bit products: (c+1)*(a+1) reaches 2^16 in a lane before the -1, which
     * is exact modulo 2^32 but would not survive a 24-bit multiply */
    lo = (((lo + 0x00010001u) * m - 0x00010001u) >> 8) & PAIR_MASK;
    hi = (((hi + 0x00010001u) * m - 0x00010001u) >> 8) & PAIR_MASK;
    const uint32_t r = lo | (hi << 8);
    const uint32_t am = 0xffu << (8 * ain);
    return (r & ~am) | (w & am);
}

/* 4 bytes starting at an arbitrary LDS byte offset, through 4-byte aligned reads */
__device__ __forceinline__ uint32_t lds_load_unaligned (const uint8_t *sb, uint32_t off)
{
    /* (byte-addressed ds_read_b32 / b64 work on gfx950 but are far slower: DESIGN.md 4.2) */
    const uint32_t *pp = reinterpret_cast<const uint32_t *> (sb + (off & ~3u));
    return __builtin_amdgcn_alignbyte (pp [1], pp [0], off & 3u);
}

/* unassociated word -> premultiplied Px2 directly, in packed 16-bit lanes:
 * (c+1)*(a+1) reaches 2^16 only as 256*256, which wraps to 0 and the -1 then
 * gives 0xffff -- the same value modulo 2^16 as the exact arithmetic. */
__device__ __forceinline__ Px2 premul_px2 (uint32_t w, uint32_t ain)
{
    const uint32_t a = (w >> (8 * ain)) & 0xffu;
    const unsigned short m = (unsigned short) (a + 1u);
    const u16x2 M2 = { m, m }, ONE2 = { 1, 1 };
    Px2 r;
    r.lo = as_u32 (((as_u16x2 (w & PAIR_MASK) + ONE2) * M2 - ONE2) >> 8);
    r.hi = as_u32 (((as_u16x2 ((w >> 8) & PAIR_MASK) + ONE2) * M2 - ONE2) >> 8);
    const uint32_t sh = 16u * (ain >> 1);
    if (ain & 1u)
        r.hi = (r.hi & ~(0xffu << sh)) | (a << sh);
    else
        r.lo = (r.lo & ~(0xffu << sh)) | (a << sh);
    return r;
}

/* How a lane reads its taps from the staged row */
enum : int
{
    TAP_A4  = 0,    /* 32-bit pixels on 4-byte LDS boundaries, already premultiplied */
    TAP_U4  = 1,    /* 32-bit pixels at any byte offset; premultiply here if needed  */
    TAP_U3  = 2     /* 24-bit pixels at any byte offset                              */
};

/* How an output pixel is stored */
enum : int
{
    OUT_P4 = 0,     /* 32-bit premultiplied, dword-aligned destination  */
    OUT_U4 = 1,     /* 32-bit unassociated, dword-aligned destination   */
    OUT_ANY = 2     /* 24-bit, or byte-aligned 32-bit (either alpha)    */
};

/* The raw LDS words of one tap: issued early (the next row's, or a batch of box pixels) so that several
 * reads are in flight, unpacked when the arithmetic wants them. */
struct TapRaw { uint32_t d0, d1, d2; };

/* rowb: LDS row base (already offset by the row's misalignment), off: byte offset of p inside the span,
 * PAIR: q is the next pixel (bilinear) or p itself (COPY / ONE / box). */
template <int TAP, bool PAIR>
__device__ __forceinline__ TapRaw tile_tap_load (const uint8_t *rowb, uint32_t off)
{
    TapRaw r;

    if (TAP == TAP_A4)
    {
        const uint32_t *pp = reinterpret_cast<const uint32_t *> (rowb + off);
        r.d0 = pp [0];
        r.d1 = PAIR ? pp [1] : 0u;
        r.d2 = 0u;
    }
    else
    {
        /* 4-byte aligned reads around the pixel(s); v_alignbyte does the rest in tile_tap_unpack */
        const uint32_t *pp = reinterpret_cast<const uint32_t *> (rowb + (off & ~3u));
        r.d0 = pp [0];
        r.d1 = pp [1];
        r.d2 = PAIR ? pp [2] : 0u;
    }
    return r;
}

/* off: the byte offset the words were loaded for (only its low two bits matter here) */
template <int TAP, bool PAIR>
__device__ __forceinline__ void tile_tap_unpack (const JobDev &J, const TapRaw &r, uint32_t off, Px2 &p, Px2 &q)
{
    uint32_t wp, wq;

    if (TAP == TAP_A4)
    {
        wp = r.d0;
        wq = PAIR ? r.d1 : wp;
    }
    else if (TAP == TAP_U4)
    {
        /* v_alignbyte shifts by the low two bits of its third operand: `off' goes in as is */
        wp = __builtin_amdgcn_alignbyte (r.d1, r.d0, off);
        if (PAIR)
            wq = __builtin_amdgcn_alignbyte (r.d2, r.d1, off);
        else
            wq = wp;
        if (J.unassoc_in)
        {
            p = premul_px2 (wp, J.pos_in [3]);
            q = PAIR ? premul_px2 (wq, J.pos_in [3]) : p;
            return;
        }
    }
    else
    {
        const uint32_t b0 = __builtin_amdgcn_alignbyte (r.d1, r.d0, off);     /* bytes 0..3 from p */
        if (PAIR)
        {
            /* bytes 0..7 from p hold both pixels (p = bytes 0-2, q = bytes 3-5): four v_perm pick
             * the packed halves directly -- selector 0x0c = 0x00, 0x0d = 0xff (the alpha lane) */
            const uint32_t b1 = __builtin_amdgcn_alignbyte (r.d2, r.d1, off);
            p.lo = __builtin_amdgcn_perm (b1, b0, 0x0c020c00u);
            p.hi = __builtin_amdgcn_perm (b1, b0, 0x0c0d0c01u);
            q.lo = __builtin_amdgcn_perm (b1, b0, 0x0c050c03u);
            q.hi = __builtin_amdgcn_perm (b1, b0, 0x0c0d0c04u);
            return;
        }
        wp = b0 | 0xff000000u;
        wq = wp;
    }

    p = px2_from_word (wp);
    q = PAIR ? px2_from_word (wq) : p;
}

/* Pixel pair (p, q) for one tap, read and unpacked in one go */
template <int TAP, bool PAIR>
__device__ __forceinline__ void tile_read_pair (const JobDev &J, const uint8_t *rowb, uint32_t off,
                                                Px2 &p, Px2 &q)
{
    const TapRaw r = tile_tap_load<TAP, PAIR> (rowb, off);
    tile_tap_unpack<TAP, PAIR> (J, r, off, p, q);
}

/* packed pair -> destination pixel word (packers of smolscale-generic.c:758-906) */
/* inv_tab: the tile's LDS copy of the inverse-alpha table (unassociated output only). A lookup
 * in global memory would make the compiler drain every outstanding vector-memory operation
 * (s_waitcnt vmcnt(0): the next group's DMA and all stores in flight) in front of each row. */
template <int OUT>
__device__ __forceinline__ uint32_t tile_pack (const JobDev &J, Px2 v, const uint32_t *inv_tab = smol_inv_p8_lut)
{
    uint32_t w = v.lo | (v.hi << 8);             /* bytes in INPUT memory order */

    if (OUT == OUT_U4 || (OUT == OUT_ANY && J.unassoc_out))
    {
        const uint32_t ain = (J.bpp_in == 4) ? J.pos_in [3] : 3u;
        const uint32_t a = (w >> (8 * ain)) & 0xffu;
        const uint32_t inv = inv_tab [a];
        uint32_t o = a << (8 * ain);
#pragma unroll
        for (uint32_t b = 0; b < 4; b++)
        {
            if (b != ain)
                o |= ((__umul24 ((w >> (8 * b)) & 0xffu, inv) >> 13) & 0xffu) << (8 * b);
        }
        w = o;
    }

    return __builtin_amdgcn_perm (w, w, J.out_perm);
}

template <int OUT, bool NT = false>
__device__ __forceinline__ void tile_store (const JobDev &J, uint8_t *rowp, uint32_t x, Px2 v, const uint32_t *inv_tab = smol_inv_p8_lut)
{
    const uint32_t w = tile_pack<OUT> (J, v, inv_tab);

    if (OUT != OUT_ANY)
    {
        /* NT: the wave's 64 dwords are contiguous (one column per lane) and never read back:
         * non-temporal (C4 189 -> 184 us; the copy-only skeleton of tools/microbench/stagebw.hip shows
         * the same 183 -> 175 us). Strided dword stores (edge tiles of the 4-columns-per-lane
         * variants) must NOT be: partial lines written around the L2 cost ratio 1.5 127 -> 211 us. */
        if (NT && J.out_wt)
            store_out32 (reinterpret_cast<SMOL_AS_GLOBAL uint32_t *> (gptr (rowp) + (size_t) x * 4), w);
        else
            *reinterpret_cast<SMOL_AS_GLOBAL uint32_t *> (gptr (rowp) + (size_t) x * 4) = w;
    }
    else if (J.bpp_out == 4)
    {
        gbyte *p = gptr (rowp) + (size_t) x * 4;
        p [0] = (uint8_t) w; p [1] = (uint8_t) (w >> 8); p [2] = (uint8_t) (w >> 16); p [3] = (uint8_t) (w >> 24);
    }
    else
    {
        gbyte *p = gptr (rowp) + (size_t) x * 3;
        p [0] = (uint8_t) w; p [1] = (uint8_t) (w >> 8); p [2] = (uint8_t) (w >> 16);
    }
}

#ifndef SMOL_KERNEL_EMU
/* memory -> LDS, 16 bytes per lane, no VGPR destination: LDS address = wave-uniform
 * base + 16 * lane (cdna_hip_programming.md section 5, global_load_lds) */
/* NON-TEMPORAL (cache policy bits = 2): the staged rows are a once-read stream; with the default policy
 * the same kernels run 1-6 % slower (BOX 115 -> 113.5 us, ratio 1.5 127 -> 124, photo thumbnails
 * 142 -> 133; profiles/r02m_dma_policy.txt), as plain loads do in k_pow2 (76 % vs 86 % of the HBM peak) */
__device__ __forceinline__ void smol_global_load_lds16 (const void *gptr, uint8_t *lds_wave_base, uint32_t)
{
    __builtin_amdgcn_global_load_lds (
        reinterpret_cast<const __attribute__ ((address_space (1))) void *> (reinterpret_cast<uintptr_t> (gptr)),
        (__attribute__ ((address_space (3))) void *) lds_wave_base, 16, 0, 2);
}

__device__ __forceinline__ void smol_wait_vmcnt0 ()
{
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
}

/* all but the wave's N youngest vector-memory operations are done (loads, LDS-DMA and stores count together, in issue
 * order: MI355X_MICROARCH.md, "s_waitcnt vmcnt(N)") */
template <int N>
__device__ __forceinline__ void smol_wait_vmcnt ()
{
    asm volatile ("s_waitcnt vmcnt(%0)" :: "n" (N) : "memory");
}

/* issue priority of this wave among the waves of its SIMD (0-3): raised while a wave requests its next row group, so
 * that the request is not queued behind the arithmetic of the waves next door */
template <int P>
__device__ __forceinline__ void smol_setprio ()
{
#if SMOL_TILE_PRIO
    if (P == 0) asm volatile ("s_setprio 0" ::: "memory");
    else asm volatile ("s_setprio 3" ::: "memory");
#endif
}

/* index of this wave in its workgroup, as a SCALAR (threadIdx.x >> 6 is wave-uniform, but the compiler keeps it
 * in a vector register, and with it every LDS address and DMA base derived from it) */
__device__ __forceinline__ uint32_t smol_wave_index ()
{
    return (uint32_t) __builtin_amdgcn_readfirstlane ((int) (threadIdx.x >> 6));
}

/* ordering point for a one-wave workgroup: compiler barrier only */
__device__ __forceinline__ void smol_wave_sync ()
{
    asm volatile ("" ::: "memory");
}

/* value of `v' in lane `idx' (idx wave-uniform) */
__device__ __forceinline__ uint32_t smol_readlane (uint32_t v, uint32_t idx)
{
    return (uint32_t) __builtin_amdgcn_readlane ((int) v, (int) idx);
}

/* value of `v' in lane `src' (src may differ per lane) */
__device__ __forceinline__ uint32_t smol_shfl_idx (uint32_t v, uint32_t src)
{
    return (uint32_t) __shfl ((int) v, (int) src, 64);
}

/* butterfly exchange inside the wave */
__device__ __forceinline__ uint32_t smol_shfl_xor (uint32_t v, uint32_t mask)
{
    return (uint32_t) __shfl_xor ((int) v, (int) mask, 64);
}
#endif

/* CPL adjacent output pixels of one lane. Fast forms: one 16-byte store of four
 * 32-bit pixels; for 24-bit output on a 4-byte aligned destination, 12 bytes per
 * lane (4 pixels per lane) or -- one pixel per lane -- the wave's 64 pixels
 * re-cut into 48 dwords through two in-wave shuffles. `interior'/`full' are
 * wave-uniform. */

template <int OUT, int CPL, bool STREAM = false>       /* STREAM: write-through whatever the rows' alignment (k_up) */
__device__ __forceinline__ void tile_store_cols (const JobDev &J, uint8_t *rowp, uint32_t xl, uint32_t xn,
                                                 bool interior, bool full24, const Px2 (&o) [CPL],
                                                 const uint32_t *inv_tab = smol_inv_p8_lut)
{
    if (CPL == 4 && OUT != OUT_ANY)
    {
        /* a full-width tile on a 16-byte aligned destination stores without
         * touching the exec mask */
        if (interior || (J.out_vec16_ok && xl + 4 <= xn))
        {
            uint32_t w [4];
#pragma unroll
            for (int c = 0; c < 4; c++)
                w [c] = tile_pack<OUT> (J, o [c], inv_tab);
            u32x4 v = { w [0], w [1], w [2], w [3] };
            if (STREAM || J.out_wt)
                store_out128 (gptr (rowp) + (size_t) xl * 4, v);
            else
            {
                /* rows that start anywhere: write-back, so that the two parts of a shared line meet in the L2 (fill_jobdev) */
                /* (k_up's rule -- whole lines non-temporal, only the lanes on the run's partial first / last line write-back -- was
                 * tried here too: sweep points 3840x2160 -> 3724, 2803, 4185 wide -3 ... -5 %, -> 3494, 6758, 7411 wide +2 ... +6 %:
                 * profiles/r05_up_handover.txt, section 5. Not kept.) */
#ifdef SMOL_KERNEL_EMU
                __builtin_memcpy (gptr (rowp) + (size_t) xl * 4, &v, 16);
#else
                *reinterpret_cast<SMOL_AS_GLOBAL u32x4a4 *> (gptr (rowp) + (size_t) xl * 4) = v;
#endif
            }
            return;
        }
    }
    if (CPL == 2 && OUT != OUT_ANY)
    {
        /* two columns per lane (round 5): the pair as one 8-byte store on any dword-aligned row */
        if (xl + 2 <= xn)
        {
            const u32x2 v = { tile_pack<OUT> (J, o [0], inv_tab), tile_pack<OUT> (J, o [1 % CPL], inv_tab) };
#ifdef SMOL_KERNEL_EMU
            __builtin_memcpy (gptr (rowp) + (size_t) xl * 4, &v, 8);
#else
            *reinterpret_cast<SMOL_AS_GLOBAL u32x2a4 *> (gptr (rowp) + (size_t) xl * 4) = v;
#endif
            return;
        }
    }
    if (OUT == OUT_ANY && full24 && CPL != 2)
    {
        /* A full-width tile of 24-bit pixels is one run of 64 * CPL * 3 bytes. It leaves as whole ALIGNED dwords
         * whatever byte the row starts on: m = (address of the run) & 3 is the same for every lane, the lanes cut the
         * run into dwords of the aligned stream -- each dword is 4 bytes of the run starting m bytes early, i.e. a
         * lane's bytes preceded by the last m bytes of the lane below it -- and only the first 4 - m and the last m
         * bytes of the run go out as single bytes (lanes 0 and 63). Rounds 1-2 stored 3 bytes per pixel one by one
         * whenever the row was not 4-byte aligned: every output width that is not a multiple of 4. */
        const uint32_t j = threadIdx.x & 63u;          /* (a no-op in one-wave workgroups) */
        gbyte *run = gptr (rowp) + (size_t) (xl - j * CPL) * 3;                 /* first byte of the tile's run */
#ifdef SMOL_KERNEL_EMU
        const uint32_t m = (uint32_t) (uintptr_t) run & 3u;
#else
        const uint32_t m = (uint32_t) __builtin_amdgcn_readfirstlane ((int) ((uint32_t) (uintptr_t) run & 3u));
#endif
        SMOL_AS_GLOBAL uint32_t *aligned = reinterpret_cast<SMOL_AS_GLOBAL uint32_t *> (run - m);
        if (CPL == 4)
        {
            uint32_t w [4];
#pragma unroll
            for (int c = 0; c < 4; c++)
                w [c] = tile_pack<OUT> (J, o [c], inv_tab) & 0x00ffffffu;
            /* this lane's 12 bytes of the run */
            const uint32_t d0 = w [0] | (w [1] << 24);
            const uint32_t d1 = (w [1] >> 8) | (w [2] << 16);
            const uint32_t d2 = (w [2] >> 16) | (w [3] << 8);
            SMOL_AS_GLOBAL uint32_t *d3 = aligned + j * 3u;
            if (m == 0)
            {
                /* three consecutive dwords (the backend merges them into one dwordx3 store) */
                d3 [0] = d0;
                d3 [1] = d1;
                d3 [2] = d2;
                return;
            }
            const uint32_t prev2 = smol_shfl_idx (d2, (j + 63u) & 63u);        /* the lane below's last dword */
            const uint32_t sh = 4u - m;
            const uint32_t t0 = __builtin_amdgcn_alignbyte (d0, prev2, sh);
            const uint32_t t1 = __builtin_amdgcn_alignbyte (d1, d0, sh);
            const uint32_t t2 = __builtin_amdgcn_alignbyte (d2, d1, sh);
            if (j != 0)
                d3 [0] = t0;
            d3 [1] = t1;
            d3 [2] = t2;
            if (j == 0)
            {
                /* the run's first 4 - m bytes */
                for (uint32_t b = 0; b < sh; b++)
                    run [b] = (uint8_t) (d0 >> (8u * b));
            }
            if (j == 63u)
            {
                /* ... and its last m bytes: the top of this lane's last dword */
                gbyte *tail = reinterpret_cast<gbyte *> (aligned + 192u);
                for (uint32_t b = 0; b < m; b++)
                    tail [b] = (uint8_t) (d2 >> (8u * (sh + b)));
            }
        }
        else
        {
            /* one pixel per lane: dword k of the aligned stream = bytes 4k - m .. 4k - m + 3 of the 192-byte run;
             * byte B of the run sits in pixel B / 3 at byte B % 3. Lanes 0..48 each build one dword from the (at most
             * two) pixels it touches, fetched with two in-wave shuffles. */
            const uint32_t w = tile_pack<OUT> (J, o [0], inv_tab) & 0x00ffffffu;
            const uint32_t B = j * 4u + 3u - m;                         /* run offset of the dword's LAST byte: never negative */
            const uint32_t la = (B < 192u ? B : 191u) / 3u;             /* pixel of the last byte  */
            const uint32_t fa = (B >= 3u ? B - 3u : 0u) / 3u;           /* pixel of the first byte (clamped for lane 0) */
            const uint32_t wa = smol_shfl_idx (w, fa & 63u);
            const uint32_t wb = smol_shfl_idx (w, la & 63u);
            /* the 6 bytes of pixels fa, fa + 1 as one 64-bit value; the dword starts at byte (B - 3) - 3 * fa of it */
            const uint32_t second = (la != fa) ? wb : 0u;
            const uint64_t six = (uint64_t) wa | ((uint64_t) second << 24);
            const int32_t first = (int32_t) B - 3;                      /* run offset of the dword's first byte (lane 0: negative) */
            if (m == 0)
            {
                if (j < 48u)
                    aligned [j] = (uint32_t) (six >> (8u * ((uint32_t) first - 3u * fa)));
                return;
            }
            if (j >= 1u && j < 48u)
                aligned [j] = (uint32_t) (six >> (8u * ((uint32_t) first - 3u * fa)));
            if (j == 0)
            {
                /* the run's first 4 - m bytes: bytes 0 .. of pixels 0 (and 1) */
                const uint32_t sh = 4u - m;
                for (uint32_t b = 0; b < sh; b++)
                    run [b] = (uint8_t) (six >> (8u * b));
            }
            if (j == 48u)
            {
                /* its last m bytes: run bytes 192 - m .. 191, the tail of pixel 63 (m <= 3 bytes: one pixel) */
                gbyte *tail = reinterpret_cast<gbyte *> (aligned + 48u);
                const uint32_t w63 = wb;                                /* la == 63 for this lane */
                for (uint32_t b = 0; b < m; b++)
                    tail [b] = (uint8_t) (w63 >> (8u * (3u - m + b)));
            }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < CPL; c++)
        if (xl + c < xn)
            tile_store<OUT, CPL == 1> (J, rowp, xl + c, o [c], inv_tab);
}


/* vertical sample schedule: (row of the top tap, weight) for sample index jy */
__device__ __forceinline__ void tile_vsample (const JobDev &J, uint32_t jy, uint32_t &oy, uint32_t &F)
{
    if (J.filter_v == F_BILINEAR)
    {
        bilin_tap (J.frac0_y + (uint64_t) jy * J.step_y, J.h_in, oy, F);
    }
    else
    {
        oy = (J.filter_v == F_ONE) ? 0u : jy;
        F = 256u;
    }
}

/* How source rows reach LDS */
enum : int
{
    TILE_STAGE_DMA = 0,             /* global_load_lds, double buffered                                   */
    TILE_STAGE_REG_PREMUL = 1,      /* through registers, premultiplying on the way (re-used pixels)      */
    TILE_STAGE_DMA_PREMUL = 2       /* DMA, then one in-place premultiply pass over the group in LDS      */
};

/* unassociated -> premultiplied on a whole pixel word in packed 16-bit lanes: (c + 1) * (a + 1) - 1 == c * (a + 1) + a
 * <= 65535, one v_pk_mad_u16 per pair of channels; asel = v_perm selector that splats the alpha byte into both lanes,
 * amask = 0xff at the alpha byte (smolscale-generic.c:238-244) */
__device__ __forceinline__ uint32_t premul_word_pk (uint32_t w, uint32_t asel, uint32_t amask)
{
    const uint32_t a2 = __builtin_amdgcn_perm (w, w, asel);              /* a | a << 16 */
    const u16x2 A2 = as_u16x2 (a2), M2 = as_u16x2 (a2 + 0x00010001u);
    const uint32_t lo = as_u32 ((as_u16x2 (w & PAIR_MASK) * M2 + A2) >> 8);
    const uint32_t hi = as_u32 ((as_u16x2 (__builtin_amdgcn_perm (w, w, 0x0c030c01u)) * M2 + A2) >> 8);
    const uint32_t r = lo | (hi << 8);
    return (r & ~amask) | (w & amask);
}

/* In-place unassociated -> premultiplied over a staged row group (32-bit pixels on 4-byte
 * boundaries, so every dword of the 16-byte chunks is one pixel): each source pixel is
 * premultiplied once instead of once per tap that touches it (1.6 times at ratio 2.5), and the
 * taps read plain aligned dwords. One wave: LDS operations complete in order, no barrier. */
__device__ __forceinline__ void tile_premul_lds (const JobDev &J, uint8_t *buf, uint32_t lane, uint32_t nrows)
{
    const uint32_t total = nrows * J.span_chunks;
    const uint32_t apos = J.pos_in [3] & 3u;
    const uint32_t asel = 0x0c000c00u | apos | (apos << 16), amask = 0xffu << (8u * apos);

    for (uint32_t c0 = 0; c0 < total; c0 += 2u * TILE_COLS)
    {
        u32x4 v [2];
#pragma unroll
        for (uint32_t u = 0; u < 2; u++)
        {
            const uint32_t c = c0 + u * TILE_COLS + lane;
            if (c < total)
                v [u] = *reinterpret_cast<const u32x4 *> (buf + (size_t) c * 16);
        }
#pragma unroll
        for (uint32_t u = 0; u < 2; u++)
        {
            const uint32_t c = c0 + u * TILE_COLS + lane;
            if (c < total)
            {
                v [u].x = premul_word_pk (v [u].x, asel, amask);
                v [u].y = premul_word_pk (v [u].y, asel, amask);
                v [u].z = premul_word_pk (v [u].z, asel, amask);
                v [u].w = premul_word_pk (v [u].w, asel, amask);
                *reinterpret_cast<u32x4 *> (buf + (size_t) c * 16) = v [u];
            }
        }
    }
}

constexpr int TILE_STAGE_UNROLL = 8;

/* Bring source rows [gfirst, gfirst + nrows) of the tile's column span into the
 * LDS buffer `buf' as 16-byte chunks (chunk c of the group lands at buf + 16*c,
 * row i starts at chunk i * span_chunks).
 *   DMA = true : global_load_lds_dwordx4 -- memory -> LDS without passing through
 *                registers; asynchronous, retired by the caller's vmcnt(0) wait.
 *   DMA = false: through registers, optionally premultiplying 32-bit unassociated
 *                pixels on the way (PREMUL). */
template <bool DMA, bool PREMUL>
__device__ __forceinline__ void tile_stage (const JobDev &J, uint8_t *buf, uint32_t lane,
                                            uint32_t gfirst, uint32_t nrows, uint32_t g0, uint32_t g1)
{
    const uintptr_t src_base = (uintptr_t) J.src;
    const size_t src_stride = J.src_stride;

    if (DMA)
    {
        /* Row by row with a running, wave-uniform row address: per row a handful of scalar
         * operations (the CU's one scalar unit is shared by all its waves and is the busiest pipe of
         * this kernel: profiles/r02p_tile_ablation.txt), the lanes only contribute the constant
         * 16 * lane as the 32-bit offset of a scalar-base load. */
        const uintptr_t ra = src_base + (size_t) gfirst * src_stride + g0;
        const uint8_t *a0 = reinterpret_cast<const uint8_t *> (ra & ~(uintptr_t) 15);  /* aligned start of the row's span */
        uint32_t mis = (uint32_t) ra & 15u;                                            /* and the span's offset from it  */
        const uint32_t span15 = g1 - g0 + 15u;
        const uint32_t pitch = J.span_chunks * 16u;
        const uint32_t lane16 = lane * 16u;
        uint8_t *rowdst = buf;

        if (J.span_chunks <= TILE_COLS)
        {
            /* the usual case: a row's span is one DMA instruction */
            for (uint32_t i = 0; i < nrows; i++)
            {
                const uint32_t nbytes = (mis + span15) & ~15u;  /* whole 16-byte chunks from the aligned start */

                if (lane16 < nbytes)
                    smol_global_load_lds16 (a0 + lane16, rowdst, lane);
                const uint32_t t = mis + (uint32_t) src_stride;
                a0 += t & ~15u;
                mis = t & 15u;
                rowdst += pitch;
            }
            return;
        }
        for (uint32_t i = 0; i < nrows; i++)
        {
            const uint32_t nbytes = (mis + span15) & ~15u;
            uint32_t k0 = 0;

            do
            {
                if (k0 + lane16 < nbytes)
                    smol_global_load_lds16 (a0 + (k0 + lane16), rowdst + k0, lane);
                k0 += TILE_COLS * 16u;
            }
            while (k0 < nbytes);
            const uint32_t t = mis + (uint32_t) src_stride;
            a0 += t & ~15u;
            mis = t & 15u;
            rowdst += pitch;
        }
        return;
    }

    const uint32_t total = nrows * J.span_chunks;
    const uint32_t ain = J.pos_in [3];

    for (uint32_t c0 = 0; c0 < total; c0 += TILE_COLS * TILE_STAGE_UNROLL)
    {
        u32x4 v [TILE_STAGE_UNROLL];
        bool ok [TILE_STAGE_UNROLL];
#pragma unroll
        for (int u = 0; u < TILE_STAGE_UNROLL; u++)
        {
            const uint32_t c = c0 + (uint32_t) u * TILE_COLS + lane;
            const uint32_t i = (uint32_t) (((uint64_t) c * J.span_chunks_inv) >> 32);
            const uint32_t k = c - i * J.span_chunks;
            const uintptr_t rb = src_base + (size_t) (gfirst + i) * src_stride;
            const uintptr_t a0 = (rb + g0) & ~(uintptr_t) 15;
            const uint32_t nch = (uint32_t) ((rb + g1 - a0 + 15) >> 4);

            ok [u] = (c < total) && (k < nch);
            if (ok [u])
                v [u] = *reinterpret_cast<const SMOL_AS_GLOBAL u32x4 *> (a0 + (uintptr_t) k * 16);
        }
#pragma unroll
        for (int u = 0; u < TILE_STAGE_UNROLL; u++)
        {
            const uint32_t c = c0 + (uint32_t) u * TILE_COLS + lane;
            if (ok [u])
            {
                if (PREMUL)
                {
                    v [u].x = premul_word (v [u].x, ain);
                    v [u].y = premul_word (v [u].y, ain);
                    v [u].z = premul_word (v [u].z, ain);
                    v [u].w = premul_word (v [u].w, ain);
                }
                *reinterpret_cast<u32x4 *> (buf + (size_t) c * 16) = v [u];
            }
        }
    }
}

/* entry points that may run several waves per workgroup (smol_k_tile.inc.h: tile_wg_shape): plain 32-bit
 * source, one sample and one column per lane, bilinear vertical filter, DMA staging */
__host__ __device__ constexpr bool tile_multiwave_variant (int tap, int hm, int stage, int cpl, int vm)
{
    return tap == 0 /* TAP_A4 */ && stage == 0 /* TILE_STAGE_DMA */ && hm == 1 && cpl == 1 && vm == TILE_VM_PAIR;
}

/* TAP: how taps are read. HM: 0 = COPY/ONE, 1/2/3 = bilinear with 1/2/4 samples
 * per output column, 4 = box with one lane per column, 5 = box with 8 lanes per
 * column (their partial sums meet through wave shuffles; for wide spans, so that
 * 64 lanes still read consecutive LDS words). STAGE (TILE_STAGE_*): DMA; premultiply while staging (TAP_A4 only; uses
 * the register path, single LDS buffer). CPL: adjacent output columns per lane
 * (a tile is 64*CPL columns wide) -- amortises the wave's scalar bookkeeping
 * when there is little vector work per column (magnification). */
template <int TAP, int HM, int OUT, int STAGE, int CPL, int VM = TILE_VM_ANY>
__device__ __forceinline__ void tile_body (const JobDev &J, uint8_t *sbuf, const uint32_t bx, const uint32_t by, const uint32_t wg_waves = 1)
{
    constexpr bool VCOPY = (VM == TILE_VM_COPY);
    constexpr bool HBOX = (HM >= 4);
    constexpr uint32_t LPC = (HM == 5) ? 8u : 1u;            /* lanes per column (box only) */
    constexpr bool HPAIR = (HM >= 1 && HM <= 3);
    constexpr uint32_t NSX = HPAIR ? (1u << (HPAIR ? HM - 1 : 0)) : 1u;
    constexpr uint32_t HX = HPAIR ? (uint32_t) (HM - 1) : 0u;
    constexpr uint32_t BPP = (TAP == TAP_U3) ? 3u : 4u;
    constexpr uint32_t TW = TILE_COLS * CPL / LPC;
    static_assert (!HBOX || CPL == 1, "box columns are one per lane group");
    constexpr bool DMA = (STAGE != TILE_STAGE_REG_PREMUL);
    /* one staging buffer, no prefetch of the next group: a property of the entry point (smol_jobdev.h sizes the
     * LDS and the groups for it under the same condition: tile_setup) */
    constexpr bool single = (STAGE == TILE_STAGE_DMA_PREMUL)
                            || (STAGE == TILE_STAGE_DMA && TAP == TAP_A4 && HM == 1 && CPL == 1 && VM == TILE_VM_PAIR);

    /* grid = (groups of column tiles, row tiles, jobs): raster order over the tiles, no integer division at the
     * head of a workgroup. A workgroup is 1 wave or 4 waves on 4 neighbouring column tiles (smol_k_tile.inc.h);
     * every wave works alone, in its own LDS region -- the DMA variants have no barriers. */
    /* (the multi-wave shape exists only for the entry points smol_k_tile.inc.h launches that way; the others keep
     * their one-wave code to the instruction) */
    constexpr bool MW = tile_multiwave_variant (TAP, HM, STAGE, CPL, VM);
    const uint32_t lane = MW ? (threadIdx.x & 63u) : threadIdx.x;
    const uint32_t tx = MW ? bx * wg_waves + smol_wave_index () : bx;
    const uint32_t ty = by;
    if (tx * TW >= J.w_out)
        return;
    const uint32_t yrel0 = ty * J.tile_rows;

    if (yrel0 >= J.n_rows)
        return;

    const uint32_t yrel1 = (yrel0 + J.tile_rows < J.n_rows) ? yrel0 + J.tile_rows : J.n_rows;
    const uint32_t x0 = tx * TW;
    const uint32_t xn = (x0 + TW < J.w_out) ? x0 + TW : J.w_out;   /* one past last column */
    const uint32_t xl = x0 + (lane / LPC) * CPL;                   /* this lane's first column */
    const uint32_t sub = lane % LPC;                               /* position inside a column's lane group */

    /* ---- horizontal taps of this lane's columns (same for every row) ---------- */
    uint32_t tap_ofs [CPL][NSX], tap_f [CPL][NSX];
    uint32_t sx0, sx1;

    uint32_t box_n = 0, box_wl = 0, box_fr = 0, box_last = 0;     /* HBOX: interior count, edge weights */

    if (HBOX)
    {
        /* column k covers source pixels o0..o1 (horizontal box, smolscale-generic.c:
         * 1427-1479): left edge weighted wl, right edge weighted fr, interior in full */
        const uint32_t xc = (xl < xn) ? xl : xn - 1;
        uint32_t o0, o1, a, b, c2, d2;
        box_span (J.step_x, J.w_in, xc, o0, o1, box_fr, box_wl);
        box_n = o1 - o0 - 1;
        tap_ofs [0][0] = o0;
        tap_f [0][0] = 0;
        box_last = o1;
        box_span (J.step_x, J.w_in, x0, sx0, a, b, c2);
        box_span (J.step_x, J.w_in, xn - 1, a, sx1, b, d2);
    }
    else if (HPAIR)
    {
#pragma unroll
        for (int c = 0; c < CPL; c++)
        {
            const uint32_t xc = (xl + c < xn) ? xl + c : xn - 1;
            uint64_t pos = J.frac0_x + (uint64_t) (xc << HX) * J.step_x;
#pragma unroll
            for (uint32_t s = 0; s < NSX; s++)
            {
                bilin_tap (pos, J.w_in, tap_ofs [c][s], tap_f [c][s]);
                pos += J.step_x;
            }
        }
        uint32_t f_unused;
        bilin_tap (J.frac0_x + (uint64_t) (x0 << HX) * J.step_x, J.w_in, sx0, f_unused);
        bilin_tap (J.frac0_x + (uint64_t) ((xn << HX) - 1) * J.step_x, J.w_in, sx1, f_unused);
        sx1 += 1;
    }
    else
    {
#pragma unroll
        for (int c = 0; c < CPL; c++)
        {
            const uint32_t xc = (xl + c < xn) ? xl + c : xn - 1;
            tap_ofs [c][0] = (J.filter_h == F_ONE) ? 0u : xc;
            tap_f [c][0] = 256u;
        }
        sx0 = (J.filter_h == F_ONE) ? 0u : x0;
        sx1 = (J.filter_h == F_ONE) ? 0u : xn - 1;
    }

    const uint32_t g0 = sx0 * BPP;                      /* first byte of the span in a row */
    const uint32_t g1 = (sx1 + 1) * BPP;                /* one past the last byte          */
    const uint32_t row_pitch = J.span_chunks * 16u;     /* LDS bytes per staged row        */
    const uint32_t buf_bytes = J.group_rows * row_pitch;
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (uint32_t s = 0; s < NSX; s++)
            tap_ofs [c][s] = (tap_ofs [c][s] - sx0) * BPP;   /* byte offset inside the span */
    box_last = (box_last - sx0) * BPP;
    /* batched box interior: pixels that need no test, and the per-lane masks of the two that do */
    const uint32_t box_fs = (uint32_t) (J.step_x >> 16);
    const uint32_t box_nfree = (HBOX && box_fs >= 2u) ? box_fs - 2u : 0u;
    /* (as values in vector registers, applied with v_and: a v_cndmask on a lane mask held in scalar registers
     * costs six times a plain vector instruction -- tools/microbench/valu_rate.hip) */
    uint32_t box_rot [4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
        box_rot [k] = BPP * ((k + lane) & 3u);
    uint32_t box_m0 = (box_nfree < box_n) ? 0xffffffffu : 0u;
    uint32_t box_m1 = (box_nfree + 1u < box_n) ? 0xffffffffu : 0u;
#ifndef SMOL_KERNEL_EMU
    asm volatile ("" : "+v" (box_m0), "+v" (box_m1));
#endif

    /* ---- vertical schedule ------------------------------------------------------
     * (source row of the top tap, weight) for every vertical sample of the tile,
     * computed once by the lanes (sample k of the tile lives in lane k & 63 of
     * sched[k >> 6]) and fetched with v_readlane in the row loop: the wave's
     * scalar unit, shared by the whole CU, stays out of the 64-bit position
     * arithmetic. */
    /* VCOPY: the vertical axis is a copy (source row = output row), known at compile time:
     * no schedule, one output row per staged row */
    /* VM: the vertical mode is a template parameter for the combinations that carry the traffic, so
     * that the row loop holds one mode's code and state; TILE_VM_ANY decides at run time */
    const bool vbox = (VM == TILE_VM_BOX) || (VM == TILE_VM_ANY && J.filter_v == F_BOX);
    const bool vpair = (VM == TILE_VM_PAIR) || (VM == TILE_VM_ANY && J.filter_v == F_BILINEAR);
    const uint32_t hv = vpair ? J.halv_v : 0u;
    const uint32_t nsy = 1u << hv;
    const uint32_t vlag = vpair ? 1u : 0u;
    uint32_t y = J.first_row + yrel0;                   /* absolute output row being built */
    const uint32_t y_end = J.first_row + yrel1;
    const uint32_t jy0 = y << hv;
    const uint32_t n_samples = (y_end - y) << hv;       /* <= TILE_MAX_VSAMPLES = 63 (tile_setup) */
    /* sample k of the tile lives in lane k; the lanes past the last sample hold a row no source row
     * index reaches (h_in <= 65535), so the row loop needs no "samples left?" test */
    constexpr uint32_t SCHED_END = 0xffffu;
    uint32_t sched = SCHED_END;
    if (!VCOPY)
    {
        uint32_t o, f;
        const uint32_t k = (lane < n_samples) ? lane : n_samples - 1u;
        if (vbox)
        {
            /* one entry per output row: last source row of its box and that row's weight */
            uint32_t o0, wl;
            box_span (J.step_y, J.h_in, jy0 + k, o0, o, f, wl);
        }
        else
        {
            tile_vsample (J, jy0 + k, o, f);
        }
        if (lane < n_samples)
            sched = o | (f << 16);
    }
    uint32_t ks = 0;                                    /* running sample index in the tile */
    uint32_t s_y = 0;                                   /* sample within the output row     */
    uint32_t e = VCOPY ? (y | (256u << 16)) : smol_readlane (sched, 0);
    uint32_t oy = e & 0xffffu, Fy = e >> 16;
    uint32_t vy0 = 0, vw1 = 0;                          /* vbox: first row of the current box, its weight */
    if (vbox)
    {
        uint32_t o1, f;
        box_span (J.step_y, J.h_in, jy0, vy0, o1, f, vw1);
    }
    const uint32_t row_first = vbox ? vy0 : oy;
    const uint32_t e_last = VCOPY ? (y_end - 1u) : smol_readlane (sched, n_samples - 1u);
    const uint32_t row_last = (e_last & 0xffffu) + vlag;
    const bool interior = (xn == x0 + TW) && J.out_vec16_ok;
    /* 24-bit output, full-width tile, 4-byte aligned rows: dword-granular stores */
    const bool full24 = (LPC == 1) && (xn == x0 + TW) && J.out3_ok;
    const uint32_t xn_store = (sub == 0) ? xn : 0u;    /* only the first lane of a column group stores */

    Px2 hprev [CPL], acc [CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++)
    {
        hprev [c].lo = hprev [c].hi = 0;
        acc [c].lo = acc [c].hi = 0;
    }
    (void) y;
    const uintptr_t src_base = (uintptr_t) J.src;
    uint8_t *out_row = J.dst + (size_t) yrel0 * J.dst_stride;

    uint32_t gfirst = row_first;
    uint32_t nrows = (row_last - gfirst + 1 < J.group_rows) ? row_last - gfirst + 1 : J.group_rows;
    uint32_t cur = 0;

#ifdef SMOL_TUNING
    const uint32_t dbg = J.dbg;         /* timing ablations: never in the shipped library */
#else
    constexpr uint32_t dbg = 0;
#endif
    const bool dbg_notaps = dbg & 1, dbg_nostage = dbg & 2, dbg_nostore = dbg & 4;

    /* unassociated output: the 1 KiB inverse-alpha table rides into LDS behind the staging buffers
     * as one 16-byte-per-lane DMA, retired by the first wait below */
    const uint32_t *inv_tab = nullptr;
    if (OUT == OUT_U4 || (OUT == OUT_ANY && J.unassoc_out))
    {
        uint8_t *tab = sbuf + ((single || (STAGE == TILE_STAGE_DMA && J.tile_ring)) ? 1u : 2u) * buf_bytes;
        smol_global_load_lds16 (reinterpret_cast<const uint8_t *> (smol_inv_p8_lut) + lane * 16u, tab, lane);
        if (!DMA)
            smol_wait_vmcnt0 ();
        inv_tab = reinterpret_cast<const uint32_t *> (tab);
    }

    /* ---- one staged source row: horizontal taps of this lane's columns, then whatever the vertical filter can emit ----
     * rowbase: the row's 16-byte aligned start in LDS; r: its index in the source image; ra_lo: low bits of the address
     * of the span's first byte in memory (its misalignment) */
    auto consume_row = [&] (const uint8_t *rowbase, const uint32_t r, const uint32_t ra_lo) __attribute__ ((always_inline))
    {
        const uint32_t mis = ra_lo & 15u;
        /* TAP_A4: mis is a multiple of 4, fold it into the row base. Otherwise keep
         * the base 16-byte aligned and add mis to each tap's byte offset, so the
         * unaligned reader's dword loads stay 4-byte aligned in LDS. */
        const uint8_t *rowb = rowbase + ((TAP == TAP_A4) ? mis : 0u);
        const uint32_t tmis = (TAP == TAP_A4) ? 0u : mis;

        Px2 h [CPL];
#pragma unroll
        for (int c = 0; c < CPL; c++)
        {
            if (dbg_notaps)
            {
                h [c].lo = tap_ofs [c][0] + r; h [c].hi = tap_f [c][0];
            }
            else if (HBOX && LPC == 1 && SMOL_TILE_BOXBATCH)
            {
                /* One lane per column: both edges and the interior in batches of four independent LDS
                 * reads. Every column of an axis has fs - 2, fs - 1 or fs interior pixels (fs = floor of
                 * the ratio; fs - 2 only where the last box is clipped at the image edge): the first
                 * fs - 2 need no test, the last two are masked per lane (box_m0 / box_m1, set up once per
                 * tile). The old loop read one pixel, waited, added: eleven LDS round trips per row. */
                Px2 pp, qq;
                const uint32_t off0 = tap_ofs [0][0] + tmis;
                const TapRaw e0 = tile_tap_load<TAP, false> (rowb, off0);
                const TapRaw e1 = tile_tap_load<TAP, false> (rowb, box_last + tmis);
                uint32_t off = off0 + BPP;
                uint32_t left = box_nfree;
                Px2 a = { 0, 0 };
                while (left >= 4u)
                {
                    /* the four pixels of a batch in a per-lane rotated order (box_rot [k] = BPP * ((k + lane) & 3)):
                     * at ratios like 64/7 the lanes 7 apart start a whole number of bank rows apart and every read
                     * of the plain order is a 5-way bank conflict (80 % of the LDS cycles of 4096 -> 448 were
                     * conflict cycles: profiles/r03c_sq_box.txt); rotated, those lanes read four different words */
                    TapRaw q4 [4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++)
                        q4 [k] = tile_tap_load<TAP, false> (rowb, off + (SMOL_TILE_BOXROT ? box_rot [k] : k * BPP));
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++)
                    {
                        tile_tap_unpack<TAP, false> (J, q4 [k], off + (SMOL_TILE_BOXROT ? box_rot [k] : k * BPP), pp, qq);
                        a = px2_add (a, pp);
                    }
                    off += 4u * BPP;
                    left -= 4u;
                }
                {
                    /* 0-3 untested pixels, then the two masked ones: one batch of up to five reads */
                    TapRaw q5 [5];
#pragma unroll
                    for (uint32_t k = 0; k < 3; k++)
                        if (k < left)
                            q5 [k] = tile_tap_load<TAP, false> (rowb, off + k * BPP);
                    const uint32_t offm = off + left * BPP;
                    q5 [3] = tile_tap_load<TAP, false> (rowb, offm);
                    q5 [4] = tile_tap_load<TAP, false> (rowb, offm + BPP);
#pragma unroll
                    for (uint32_t k = 0; k < 3; k++)
                        if (k < left)
                        {
                            tile_tap_unpack<TAP, false> (J, q5 [k], off + k * BPP, pp, qq);
                            a = px2_add (a, pp);
                        }
                    tile_tap_unpack<TAP, false> (J, q5 [3], offm, pp, qq);
                    pp.lo &= box_m0; pp.hi &= box_m0;
                    a = px2_add (a, pp);
                    tile_tap_unpack<TAP, false> (J, q5 [4], offm + BPP, pp, qq);
                    pp.lo &= box_m1; pp.hi &= box_m1;
                    a = px2_add (a, pp);
                }
                /* the edges were requested first and are consumed last */
                tile_tap_unpack<TAP, false> (J, e0, off0, pp, qq);
                a = px2_add (a, px2_mulw (pp, box_wl));
                tile_tap_unpack<TAP, false> (J, e1, box_last + tmis, pp, qq);
                a = px2_add (a, px2_mulw (pp, box_fr));
                h [c] = px2_box_scale (a, J.span_mul_x);
            }
            else if (HBOX)
            {
                Px2 pp, qq;
                Px2 a = { 0, 0 };
                if (sub == 0)
                {
                    tile_read_pair<TAP, false> (J, rowb, tap_ofs [0][0] + tmis, pp, qq);
                    a = px2_mulw (pp, box_wl);
                    tile_read_pair<TAP, false> (J, rowb, box_last + tmis, pp, qq);
                    a = px2_add (a, px2_mulw (pp, box_fr));
                }
                /* interior pixels: lane `sub' of the group takes every LPC-th one */
                for (uint32_t ii = sub; ii < J.box_nmax; ii += LPC)
                {
                    if (ii < box_n)
                    {
                        tile_read_pair<TAP, false> (J, rowb, tap_ofs [0][0] + tmis + (ii + 1) * BPP, pp, qq);
                        a = px2_add (a, pp);
                    }
                }
                if (LPC > 1)
                {
#pragma unroll
                    for (uint32_t m = 1; m < LPC; m <<= 1)
                    {
                        Px2 o2;
                        o2.lo = smol_shfl_xor (a.lo, m);
                        o2.hi = smol_shfl_xor (a.hi, m);
                        a = px2_add (a, o2);
                    }
                }
                h [c] = px2_box_scale (a, J.span_mul_x);
            }
            else if (HM == 0)
            {
                Px2 pp, qq;
                tile_read_pair<TAP, false> (J, rowb, tap_ofs [c][0] + tmis, pp, qq);
                h [c] = pp;
            }
            else
            {
#pragma unroll
                for (uint32_t s = 0; s < NSX; s++)
                {
                    Px2 pp, qq;
                    tile_read_pair<TAP, true> (J, rowb, tap_ofs [c][s] + tmis, pp, qq);
                    const Px2 t = tap2 (pp, qq, tap_f [c][s]);
                    h [c] = s ? px2_add (h [c], t) : t;
                }
                if (HX)
                    h [c] = px2_shr (h [c], HX);
            }
        }

        if (vbox)
        {
            /* vertical box (smolscale-generic.c:2026-2161): rows arrive in order;
             * the box's first row is weighted vw1, interior rows count in full, its
             * last row (oy) is weighted Fy and, weighted 255 - Fy, also opens the
             * next box. Sums stay below 2^16 per lane, as in the reference. */
            if (r == vy0)
                acc [0] = px2_mulw (h [0], vw1);
            else if (r < oy)
                acc [0] = px2_add (acc [0], h [0]);
            if (r == oy)
            {
                Px2 o [CPL];
                o [0] = px2_box_scale (px2_add (acc [0], px2_mulw (h [0], Fy)), J.span_mul_y);
                if (!dbg_nostore)
                    tile_store_cols<OUT, CPL> (J, out_row, xl, xn_store, false, full24, o, inv_tab);
                out_row += J.dst_stride;
                ks++;
                vy0 = oy;
                vw1 = 255u - Fy;
                acc [0] = px2_mulw (h [0], vw1);
                e = smol_readlane (sched, ks);      /* past the last sample: the sentinel */
                oy = e & 0xffffu;
                Fy = e >> 16;
            }
        }
        else if (VCOPY)
        {
            /* vertical copy: this row is an output row */
            if (!dbg_nostore)
                tile_store_cols<OUT, CPL> (J, out_row, xl, xn_store, interior && LPC == 1, full24, h, inv_tab);
            out_row += J.dst_stride;
        }
        else
        /* vertical: emit every sample whose last row is r */
        while (oy + vlag == r)
        {
            if (hv == 0)
            {
                /* no vertical halvings: the sample is the output row */
                Px2 o [CPL];
#pragma unroll
                for (int c = 0; c < CPL; c++)
                    o [c] = vpair ? tap2 (hprev [c], h [c], Fy) : h [c];
                if (!dbg_nostore)
                    tile_store_cols<OUT, CPL> (J, out_row, xl, xn_store, interior && LPC == 1, full24, o, inv_tab);
                out_row += J.dst_stride;
                ks++;
                e = smol_readlane (sched, ks);
                oy = e & 0xffffu;
                Fy = e >> 16;
                continue;
            }
            /* acc is zero at the first sample of an output row */
#pragma unroll
            for (int c = 0; c < CPL; c++)
            {
                Px2 t = h [c];
                if (vpair)
                    t = tap2 (hprev [c], h [c], Fy);
                acc [c] = px2_add (acc [c], t);
            }
            s_y++;
            ks++;
            if (s_y == nsy)
            {
                Px2 o [CPL];
#pragma unroll
                for (int c = 0; c < CPL; c++)
                {
                    o [c] = px2_shr (acc [c], hv);
                    acc [c].lo = acc [c].hi = 0;
                }
                if (!dbg_nostore)
                    tile_store_cols<OUT, CPL> (J, out_row, xl, xn_store, interior && LPC == 1, full24, o, inv_tab);
                s_y = 0;
                out_row += J.dst_stride;
            }
            e = smol_readlane (sched, ks);          /* past the last sample: the sentinel */
            oy = e & 0xffffu;
            Fy = e >> 16;
        }
        if (!VCOPY)
        {
#pragma unroll
            for (int c = 0; c < CPL; c++)
                hprev [c] = h [c];
        }
    };

    /* ---- ring staging (round 6) -------------------------------------------------------------------------------------
     * The two-buffer scheme keeps ONE row group in flight per wave while the other is consumed, and a tile of 13 source rows
     * pays the memory latency two or three times in a row. Here the LDS holds a ring of J.group_rows rows: the tile's first
     * rows are all requested up front, a row is consumed as soon as IT has landed -- vector-memory operations retire in
     * issue order, so "all but the N youngest are done" (s_waitcnt vmcnt(N)) means row r is there once the ring's other
     * rows, at least N DMA instructions, were requested after it -- and its slot is handed to the next row at once: the
     * ring's other rows stay in flight through the arithmetic. The wait takes an immediate, so N = J.tile_ring is one of a
     * few values; the host sizes the ring so that the rows behind any row make up at least N instructions (tile_setup),
     * tiles whose rows are shorter than that (clipped at the image's edge) drain instead. */
    if (DMA && STAGE == TILE_STAGE_DMA && J.tile_ring)
    {
        const uint32_t ring_n = J.tile_ring;                    /* 2, 3, 4, 6, 8 or 12 (tile_setup) */
        const uint32_t ring_rows = J.group_rows;
        const uint32_t total = row_last - row_first + 1u;
        const uintptr_t ra = src_base + (size_t) row_first * J.src_stride + g0;
        const uint8_t *a0 = reinterpret_cast<const uint8_t *> (ra & ~(uintptr_t) 15);  /* aligned start of the next row to request */
        uint32_t imis = (uint32_t) ra & 15u;
        const uint32_t span15 = g1 - g0 + 15u;
        const uint32_t lane16 = lane * 16u;
        uint8_t *const ring_end = sbuf + ring_rows * row_pitch;
        uint8_t *slot = sbuf;                                   /* the slot that is consumed next; once consumed, requested into */
        uint32_t to_issue = total;
        /* DMA instructions a row of this tile takes at least (its span with no misalignment in front) */
        const uint32_t kmin = ((span15 & ~15u) + TILE_COLS * 16u - 1u) / (TILE_COLS * 16u);
        const bool counted = (ring_rows - 1u) * kmin >= ring_n;

        auto request_row = [&] (uint8_t *rowdst) __attribute__ ((always_inline))
        {
            const uint32_t nbytes = (imis + span15) & ~15u;     /* whole 16-byte chunks from the aligned start */
            if (!dbg_nostage)
            {
                uint32_t k0 = 0;
                do
                {
                    if (k0 + lane16 < nbytes)
                        smol_global_load_lds16 (a0 + (k0 + lane16), rowdst + k0, lane);
                    k0 += TILE_COLS * 16u;
                }
                while (k0 < nbytes);
            }
            const uint32_t t = imis + (uint32_t) J.src_stride;
            a0 += t & ~15u;
            imis = t & 15u;
            to_issue--;
        };

        {
            uint8_t *d = sbuf;
            smol_setprio<3> ();
            for (uint32_t n = total < ring_rows ? total : ring_rows; n; n--, d += row_pitch)
                request_row (d);
            smol_setprio<0> ();
        }
        uint32_t ra_lo = (uint32_t) ra;
        uint32_t r = row_first;
        /* steady state: one row out, one row in */
        for (; to_issue; r++, ra_lo += J.src_stride)
        {
            if (!counted)
                smol_wait_vmcnt0 ();
            else if (ring_n == 4u)              /* (what the shipped rule picks; the others: tuning builds) */
                smol_wait_vmcnt<4> ();
            else if (ring_n == 8u)
                smol_wait_vmcnt<8> ();
            else if (ring_n == 6u)
                smol_wait_vmcnt<6> ();
            else if (ring_n == 12u)
                smol_wait_vmcnt<12> ();
            else if (ring_n == 3u)
                smol_wait_vmcnt<3> ();
            else
                smol_wait_vmcnt<2> ();
            smol_wave_sync ();
            consume_row (slot, r, ra_lo);
#if defined (SMOL_TUNING) && !defined (SMOL_KERNEL_EMU)
            /* sensitivity probes (profiles/r06_sq_tile_classes.txt): sixteen more scalar (8) or vector (16) instructions per source row */
            if (dbg & 8)
            {
                uint32_t pad = r;
                asm volatile ("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\t"
                              "s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\t"
                              "s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\t"
                              "s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1" : "+s" (pad) : : "scc");
            }
            if (dbg & 16)
            {
                uint32_t pad = lane;
                asm volatile ("v_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\t"
                              "v_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\t"
                              "v_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\t"
                              "v_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1" : "+v" (pad));
            }
#endif
            smol_wave_sync ();
            smol_setprio<3> ();
            request_row (slot);
            smol_setprio<0> ();
            slot += row_pitch;
            if (slot == ring_end)
                slot = sbuf;
        }
        /* the ring's last rows: everything is requested */
        smol_wait_vmcnt0 ();
        smol_wave_sync ();
        for (; r <= row_last; r++, ra_lo += J.src_stride)
        {
            consume_row (slot, r, ra_lo);
            slot += row_pitch;
            if (slot == ring_end)
                slot = sbuf;
        }
        return;
    }

    if (DMA && !dbg_nostage && !single)
        tile_stage<true, false> (J, sbuf, lane, gfirst, nrows, g0, g1);

    while (true)
    {
        uint8_t *buf = sbuf + cur * buf_bytes;

        if (DMA && single)
        {
            /* one buffer: request the group, wait for it (the waves next door fill the gap) */
            smol_wave_sync ();
            if (!dbg_nostage)
                tile_stage<true, false> (J, buf, lane, gfirst, nrows, g0, g1);
            smol_wait_vmcnt0 ();
            smol_wave_sync ();
            if (STAGE == TILE_STAGE_DMA_PREMUL && !dbg_notaps)
            {
                tile_premul_lds (J, buf, lane, nrows);
                smol_wave_sync ();
            }
        }
        else if (DMA)
        {
            /* Retire this group's DMA (issued before the previous group's pixel stores), then start
             * the next group into the other buffer so that it streams in during the consume.
             * A counted wait that lets the stores fly (vmcnt(n) for the n stores issued since) measured
             * the same as draining everything (profiles/r02p_wait_ab.txt) and cost scalar bookkeeping. */
#if SMOL_TILE_PRIO == 2
            smol_setprio<3> ();
#endif
            smol_wait_vmcnt0 ();
            /* the workgroup is a single wave: no s_barrier needed, and __syncthreads()
             * would make hipcc drain every outstanding store (vmcnt(0)) */
            smol_wave_sync ();
            const uint32_t nfirst = gfirst + nrows;
            if (nfirst <= row_last && !dbg_nostage)
            {
                const uint32_t nn = (row_last - nfirst + 1 < J.group_rows) ? row_last - nfirst + 1 : J.group_rows;
                smol_setprio<3> ();
                tile_stage<true, false> (J, sbuf + (cur ^ 1u) * buf_bytes, lane, nfirst, nn, g0, g1);
            }
            smol_setprio<0> ();
            if (STAGE == TILE_STAGE_DMA_PREMUL && !dbg_notaps)
            {
                tile_premul_lds (J, buf, lane, nrows);
                smol_wave_sync ();      /* (a real barrier only in the host emulation, where lanes are threads) */
            }
        }
        else
        {
            __syncthreads ();                           /* previous group fully consumed */
            if (!dbg_nostage)
                tile_stage<false, STAGE == TILE_STAGE_REG_PREMUL> (J, buf, lane, gfirst, nrows, g0, g1);
            __syncthreads ();
        }

        /* ---- consume the staged rows in order ------------------------------------- */
        uint32_t ra_lo = (uint32_t) (src_base + (size_t) gfirst * J.src_stride + g0);   /* low bits of the row's first byte */
        const uint8_t *rowbase = buf;
        for (uint32_t i = 0; i < nrows; i++, ra_lo += J.src_stride, rowbase += row_pitch)
            consume_row (rowbase, gfirst + i, ra_lo);

        gfirst += nrows;
        if (gfirst > row_last)
            break;
        nrows = (row_last - gfirst + 1 < J.group_rows) ? row_last - gfirst + 1 : J.group_rows;
        if (!single)
            cur ^= 1u;
    }
}

template <int TAP, int HM, int STAGE, int CPL, int VM = TILE_VM_ANY>
__device__ __forceinline__ void tile_dispatch_out (const JobDev &J, uint8_t *sbuf, const uint32_t bx, const uint32_t by, const uint32_t wg_waves = 1)
{
    switch (J.tile_variant & 3u)
    {
        case OUT_P4: tile_body<TAP, HM, OUT_P4, STAGE, CPL, VM> (J, sbuf, bx, by, wg_waves); break;
        case OUT_U4: tile_body<TAP, HM, OUT_U4, STAGE, CPL, VM> (J, sbuf, bx, by, wg_waves); break;
        default:     tile_body<TAP, HM, OUT_ANY, STAGE, CPL, VM> (J, sbuf, bx, by, wg_waves); break;
    }
}

template <int TAP, int STAGE>
__device__ __forceinline__ void tile_dispatch_hm (const JobDev &J, uint8_t *sbuf)
{
    const uint32_t hm = (J.tile_variant >> 2) & 7u;

    /* (host emulation only: the GPU build has one entry point per combination) */
    if (J.tile_cpl == 2)
    {
        switch (hm)
        {
            case 1:  tile_dispatch_out<TAP, 1, STAGE, 2, TILE_VM_PAIR> (J, sbuf, blockIdx.x, blockIdx.y); break;
            case 2:  tile_dispatch_out<TAP, 2, STAGE, 2, TILE_VM_PAIR> (J, sbuf, blockIdx.x, blockIdx.y); break;
            default: tile_dispatch_out<TAP, 3, STAGE, 2, TILE_VM_PAIR> (J, sbuf, blockIdx.x, blockIdx.y); break;
        }
        return;
    }
    if (J.tile_cpl == 4)
    {
        if (J.tile_vm == TILE_VM_PAIR)
            switch (hm)
            {
                case 1:  tile_dispatch_out<TAP, 1, STAGE, 4, TILE_VM_PAIR> (J, sbuf, blockIdx.x, blockIdx.y); break;
                case 2:  tile_dispatch_out<TAP, 2, STAGE, 4, TILE_VM_PAIR> (J, sbuf, blockIdx.x, blockIdx.y); break;
                default: tile_dispatch_out<TAP, 3, STAGE, 4, TILE_VM_PAIR> (J, sbuf, blockIdx.x, blockIdx.y); break;
            }
        else
            switch (hm)
            {
                case 0:  tile_dispatch_out<TAP, 0, STAGE, 4> (J, sbuf, blockIdx.x, blockIdx.y); break;
                case 1:  tile_dispatch_out<TAP, 1, STAGE, 4> (J, sbuf, blockIdx.x, blockIdx.y); break;
                case 2:  tile_dispatch_out<TAP, 2, STAGE, 4> (J, sbuf, blockIdx.x, blockIdx.y); break;
                default: tile_dispatch_out<TAP, 3, STAGE, 4> (J, sbuf, blockIdx.x, blockIdx.y); break;
            }
        return;
    }

    if (J.tile_vm == TILE_VM_COPY)
    {
        switch (hm)
        {
            case 1:  tile_dispatch_out<TAP, 1, STAGE, 1, TILE_VM_COPY> (J, sbuf, blockIdx.x, blockIdx.y); break;
            case 2:  tile_dispatch_out<TAP, 2, STAGE, 1, TILE_VM_COPY> (J, sbuf, blockIdx.x, blockIdx.y); break;
            default: tile_dispatch_out<TAP, 3, STAGE, 1, TILE_VM_COPY> (J, sbuf, blockIdx.x, blockIdx.y); break;
        }
        return;
    }
    if (J.tile_vm == TILE_VM_PAIR)
    {
        switch (hm)
        {
            case 1:  tile_dispatch_out<TAP, 1, STAGE, 1, TILE_VM_PAIR> (J, sbuf, blockIdx.x, blockIdx.y); break;
            case 2:  tile_dispatch_out<TAP, 2, STAGE, 1, TILE_VM_PAIR> (J, sbuf, blockIdx.x, blockIdx.y); break;
            default: tile_dispatch_out<TAP, 3, STAGE, 1, TILE_VM_PAIR> (J, sbuf, blockIdx.x, blockIdx.y); break;
        }
        return;
    }
    if (J.tile_vm == TILE_VM_BOX)
    {
        if (hm == 4)
            tile_dispatch_out<TAP, 4, STAGE, 1, TILE_VM_BOX> (J, sbuf, blockIdx.x, blockIdx.y);
        else
            tile_dispatch_out<TAP, 5, STAGE, 1, TILE_VM_BOX> (J, sbuf, blockIdx.x, blockIdx.y);
        return;
    }

    switch (hm)
    {
        case 0:  tile_dispatch_out<TAP, 0, STAGE, 1> (J, sbuf, blockIdx.x, blockIdx.y); break;
        case 1:  tile_dispatch_out<TAP, 1, STAGE, 1> (J, sbuf, blockIdx.x, blockIdx.y); break;
        case 2:  tile_dispatch_out<TAP, 2, STAGE, 1> (J, sbuf, blockIdx.x, blockIdx.y); break;
        case 3:  tile_dispatch_out<TAP, 3, STAGE, 1> (J, sbuf, blockIdx.x, blockIdx.y); break;
        case 4:  tile_dispatch_out<TAP, 4, STAGE, 1> (J, sbuf, blockIdx.x, blockIdx.y); break;
        default: tile_dispatch_out<TAP, 5, STAGE, 1> (J, sbuf, blockIdx.x, blockIdx.y); break;
    }
}

/* tile_variant = tapsel << 5 | HM << 2 | OUT; tapsel: 0 A4, 1 A4 + premultiply
 * while staging, 2 U4, 3 U3, 4 A4 after an in-LDS premultiply pass (smol_jobdev.h: tile_setup) */
__device__ __forceinline__ void tile_dispatch (const JobDev &J, uint8_t *sbuf)
{
    switch (J.tile_variant >> 5)
    {
        case 0:  tile_dispatch_hm<TAP_A4, TILE_STAGE_DMA> (J, sbuf); break;
        case 1:  tile_dispatch_hm<TAP_A4, TILE_STAGE_REG_PREMUL> (J, sbuf); break;
        case 2:  tile_dispatch_hm<TAP_U4, TILE_STAGE_DMA> (J, sbuf); break;
        case 3:  tile_dispatch_hm<TAP_U3, TILE_STAGE_DMA> (J, sbuf); break;
        default: tile_dispatch_hm<TAP_A4, TILE_STAGE_DMA_PREMUL> (J, sbuf); break;
    }
}

#ifndef SMOL_KERNEL_EMU
extern __shared__ __attribute__ ((aligned (16))) uint8_t smol_dyn_lds [];
#endif

/* One __global__ entry per (tap reader, horizontal mode, columns per lane): each
 * gets its own register allocation, so a lean variant (a 1-tap copy) is not held
 * to the occupancy of the heaviest one (4 columns x 4 samples). TS = tapsel of
 * tile_variant: 0 A4, 1 A4 + premultiply while staging, 2 U4, 3 U3. */
template <int TS, int HM, int CPL, int VM = TILE_VM_ANY>
__device__ __forceinline__ void tile_entry (const JobDev &J, uint8_t *sbuf, const uint32_t bx, const uint32_t by, const uint32_t wg_waves = 1)
{
    constexpr int TAP = (TS <= 1 || TS == 4) ? TAP_A4 : (TS == 2 ? TAP_U4 : TAP_U3);
    constexpr int STAGE = (TS == 1) ? TILE_STAGE_REG_PREMUL : (TS == 4 ? TILE_STAGE_DMA_PREMUL : TILE_STAGE_DMA);
    tile_dispatch_out<TAP, HM, STAGE, CPL, VM> (J, sbuf, bx, by, wg_waves);
}

#ifdef SMOL_KERNEL_EMU
/* host emulation: one entry that switches at run time */
__global__ void k_tile_one (const JobDev J)
{
    tile_dispatch (J, smol_dyn_lds);
}
#else
template <int TS, int HM, int CPL, int VM = TILE_VM_ANY>
__global__ void __launch_bounds__ (tile_multiwave_variant (TS == 0 ? 0 : 1, HM, TS == 0 ? 0 : 1, CPL, VM) ? 256 : 64)
k_tile_batch (const JobDev *__restrict__ jobs, const uint32_t lds_per_wave, const uint32_t wg_waves, const TileSwz swz)
{
    uint32_t bx, by, bz;
    tile_coords (swz, bx, by, bz);
    /* private copy: the descriptor must not be re-read from memory after every
     * pixel store (byte-typed stores may alias anything as far as the compiler knows) */
    const JobDev J = load_job (jobs, bz);
    constexpr bool MW = tile_multiwave_variant (TS == 0 ? 0 : 1, HM, TS == 0 ? 0 : 1, CPL, VM);
    tile_entry<TS, HM, CPL, VM> (J, MW ? smol_dyn_lds + smol_wave_index () * lds_per_wave : smol_dyn_lds, bx, by, wg_waves);
}

template <int TS, int HM, int CPL, int VM = TILE_VM_ANY>
__global__ void __launch_bounds__ (tile_multiwave_variant (TS == 0 ? 0 : 1, HM, TS == 0 ? 0 : 1, CPL, VM) ? 256 : 64)
k_tile_one (const JobDev J, const uint32_t lds_per_wave, const uint32_t wg_waves, const TileSwz swz)
{
    uint32_t bx, by, bz;
    tile_coords (swz, bx, by, bz);
    constexpr bool MW = tile_multiwave_variant (TS == 0 ? 0 : 1, HM, TS == 0 ? 0 : 1, CPL, VM);
    tile_entry<TS, HM, CPL, VM> (J, MW ? smol_dyn_lds + smol_wave_index () * lds_per_wave : smol_dyn_lds, bx, by, wg_waves);
}
#endif

} /* namespace smolhip */
