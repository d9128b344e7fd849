/* smol_kernel_box2.hip.h -- k_boxh + k_boxv: the box filter on both axes as TWO
 * passes through an intermediate image, for strong minification (photo ->
 * thumbnail) and for every box job of the 128bpp class. Included after
 * smol_kernel_tile.hip.h (packed-lane helpers, LDS staging, pixel stores).
 *
 * A single-pass tile kernel has one lane per output column and one wave per 64
 * columns x a few rows: at a ratio of 20 a 200 x 150 thumbnail is a few hundred
 * waves that each walk 1300 source pixels per row, far too few to fill 256 CUs.
 * Splitting the axes puts the parallelism where the work is:
 *
 *   k_boxh  one wave per 64 output columns x 1 SOURCE row (4 rows in the 128bpp class, one after the
 *           other); the 4 waves of a workgroup take 4 neighbouring column tiles of the same row(s),
 *           grid = (column-tile groups, row groups, jobs). The row arrives in the wave's ONE LDS row
 *           buffer by LDS-DMA -- no prefetch inside the wave: with a single buffer 32 waves fit a CU and
 *           cover each other's latency, which measured faster than 20 waves with two buffers each --
 *           a lane sums its column's span (smolscale-generic.c:1427-1556:
 *           left edge weighted, interior in full, right edge weighted, then the
 *           span_mul normalisation) and writes the horizontal result -- four 16-bit
 *           channels, 8 bytes -- to the intermediate image (coalesced).
 *   k_boxv  lane = output column (or 8 lanes per column for tall boxes), one wave per output row: sums the
 *           intermediate rows of the box (:2026-2260) with 8-byte coalesced loads,
 *           normalises, packs and stores the pixel.
 *
 * The intermediate image is h_rows x w_out x 8 bytes (a few per cent of the source),
 * allocated by the runtime. 64bpp class: values are 8 bit, two per register in
 * 16-bit lanes (the Px2 arithmetic of k_tile). 128bpp class: pixels go through
 * unpack_word<true> / store_px<true> with the tables in LDS, sums are 32 bit. */

#pragma once

#include "smol_kernel_lind.hip.h"

namespace smolhip {

using dispatch::twopass_lin_h;
using dispatch::twopass_lin_v;


template <bool V> struct BoxAligned { static constexpr bool value = V; };

/* One staged source pixel at byte offset `off' of the LDS row. AL (compile time, chosen once per wave outside the
 * row loop -- as a per-read test it cost the 24-bit thumbnails 4-8 %): 32-bit pixels on 4-byte boundaries, which sit
 * on 4-byte LDS boundaries too (the staged row keeps the source's phase): one aligned read (gamma-correct 4096^2 ->
 * 448^2 286 -> 271 us); everything else through two aligned dwords and v_alignbyte. */
template <bool AL>
__device__ __forceinline__ uint32_t box2_px_word (const JobDev &J, const uint8_t *rowb, uint32_t off)
{
    if (AL)
        return *reinterpret_cast<const uint32_t *> (rowb + off);
    const uint32_t w = lds_load_unaligned (rowb, off);
    return (J.bpp_in == 3) ? (w | 0xff000000u) : w;
}

/* LPC = lanes per output column: 1 (64 columns per wave), for very wide spans 8 (8 columns per wave) or -- spans of
 * hundreds of pixels, 128bpp class only -- 16 (from a ratio of 96) or 64 (from 400: the whole wave on one column): a row
 * buffer of four spans or one instead of eight, so that enough waves fit a CU to cover each other's loads; originally: 1 or, for very wide spans, 8 (8 columns
 * per wave; the 8 lanes of a column take consecutive eighths of its interior pixels and meet
 * through wave shuffles -- sums of exact integers, so the order does not matter) */
template <bool WIDE, int LPC>
__device__ __forceinline__ void boxh_body (const JobDev &J, uint8_t *lds, const uint8_t *lut_lds,
                                           const uint32_t lane, const uint32_t tx, const uint32_t tr)
{
    constexpr uint32_t COLS = TILE_COLS / LPC;
    const uint32_t sub = lane % LPC;                                /* position in the column's lane group */
    constexpr uint32_t BOXH_ROWS = WIDE ? SMOL_BOXH_ROWS_WIDE : SMOL_BOXH_ROWS;     /* source rows per wave (smol_jobdev.h: boxh_rows) */
    const uint32_t r0 = J.tmp_row0 + tr * BOXH_ROWS;                /* first source row of this wave */

    if (r0 > J.tmp_row1 || tx * COLS >= J.w_out)
        return;

    const uint32_t r1 = (r0 + BOXH_ROWS - 1 < J.tmp_row1) ? r0 + BOXH_ROWS - 1 : J.tmp_row1;
    const uint32_t x0 = tx * COLS;
    const uint32_t xn = (x0 + COLS < J.w_out) ? x0 + COLS : J.w_out;
    const uint32_t x = x0 + lane / LPC;
    const uint32_t xc = (x < xn) ? x : xn - 1;
    const uint32_t bpp = J.bpp_in;

    /* the column's span: pixels o0 (weight wl), o0+1 .. o1-1 (in full), o1 (weight fr) */
    uint32_t o0, o1, fr, wl;
    box_span (J.step_x, J.w_in, xc, o0, o1, fr, wl);
    /* the tile's span: from the first lane's first pixel to the last lane's last (lane 63 sits on column
     * xn - 1 in full and in partial tiles alike) -- two v_readlane instead of two more 64-bit box_span */
    const uint32_t sx0 = smol_readlane (o0, 0);
    const uint32_t sx1 = smol_readlane (o1, 63);
    const uint32_t n_col = o1 - o0 - 1;                             /* interior pixels of the column */
    /* this lane's share of the interior: pixels [i_begin, i_begin + n_int) of it */
    const uint32_t per = (LPC == 1) ? n_col : (n_col + LPC - 1) / LPC;
    const uint32_t i_begin = (LPC == 1) ? 0u : ((sub * per < n_col) ? sub * per : n_col);
    const uint32_t n_int = (LPC == 1) ? n_col : ((i_begin + per < n_col) ? per : n_col - i_begin);
    const bool edges = (sub == 0);                                  /* the first lane of a group adds both weighted edge pixels */
    /* 24-bit narrow path: byte masks of the last, partial group of four interior pixels */
    const uint32_t rem_px = n_int & 3u;
    const uint32_t rem_mask0 = (rem_px == 0) ? 0u : (rem_px == 1 ? 0x00ffffffu : 0xffffffffu);
    const uint32_t rem_mask1 = (rem_px <= 1) ? 0u : (rem_px == 2 ? 0x0000ffffu : 0xffffffffu);
    const uint32_t rem_mask2 = (rem_px == 3) ? 0x000000ffu : 0u;
    const uint32_t g0 = sx0 * bpp, g1 = (sx1 + 1) * bpp;
    const uint32_t first_ofs = (o0 - sx0) * bpp;
    const uint32_t int_ofs = first_ofs + (1u + i_begin) * bpp;      /* first interior pixel of this lane */
    const uint32_t last_ofs = first_ofs + (n_col + 1u) * bpp;       /* the column's last pixel */
    const LutLds lut (lut_lds, LUT_AT_LDS_0);       /* (boxh_entry / boxv_entry: lut_expect_lds_0) */
    const uintptr_t src_base = (uintptr_t) J.src;

    auto rows = [&] (auto aligned_tag, const auto &un)
    {
    constexpr bool AL = decltype (aligned_tag)::value;
    bool try_opaque = true;
    for (uint32_t r = r0; r <= r1; r++)
    {
        /* one row buffer per wave: request the row, wait for it, sum it (the waves next door fill the gap) */
        smol_wave_sync ();
        tile_stage<true, false> (J, lds, lane, r, 1, g0, g1);
        smol_wait_vmcnt0 ();
        smol_wave_sync ();

        const uint8_t *rowb = lds;
        const uint32_t mis = (uint32_t) (src_base + (size_t) r * J.src_stride + g0) & 15u;
        uint32_t off = first_ofs + mis;
        u32x2a8 res;

        if (WIDE)
        {
            uint32_t acc [4] = { 0, 0, 0, 0 }, c [4];
            if (edges)
            {
                un (box2_px_word<AL> (J, rowb, off), c);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc [i] = ((c [i] * wl) >> 8) & 0x00ffffffu;
                un (box2_px_word<AL> (J, rowb, last_ofs + mis), c);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc [i] += ((c [i] * fr) >> 8) & 0x00ffffffu;
            }
            off = int_ofs + mis;
            constexpr int UK0 = std::remove_reference_t<decltype (un)>::KIND;
            if (UK0 == WU_LIN_O && bpp == 3)
            {
                /* Gamma-correct 24-bit rows (photographs -> thumbnails): every byte of the span goes through the SAME
                 * 256-entry table (opaque pixels: one fused look-up per channel), so the interior is a stream of bytes.
                 * Groups of 4 pixels = 12 bytes = 3 dwords, realigned to the lane's first interior byte with three
                 * v_alignbyte (as the 64bpp class does for its dot products): 12 x (v_bfe, address, look-up, add) per
                 * group, no per-pixel extraction, no per-pixel select; the 0-3 pixels left over are one more group with
                 * the bytes past the span masked to 0, which the table maps to 0. The sums are kept per byte POSITION
                 * (0, 1, 2 of a pixel) and handed to the channels at the end. Gamma-correct photo thumbnails
                 * 192 -> see profiles/r04_boxh24_ab.txt. */
                const uint32_t ngrp = n_int >> 2;
                uint32_t s0 = 0, s1 = 0, s2 = 0;
                const uint32_t sh = off & 3u;
                const uint32_t *pp = reinterpret_cast<const uint32_t *> (rowb + (off & ~3u));
                for (uint32_t g = 0; g <= J.box_nmax; g += 4, pp += 3)
                {
                    if ((g >> 2) <= ngrp)
                    {
                        const uint32_t q0 = pp [0], q1 = pp [1], q2 = pp [2], q3 = pp [3];
                        uint32_t d0 = __builtin_amdgcn_alignbyte (q1, q0, sh);
                        uint32_t d1 = __builtin_amdgcn_alignbyte (q2, q1, sh);
                        uint32_t d2 = __builtin_amdgcn_alignbyte (q3, q2, sh);
                        if ((g >> 2) == ngrp)
                        {
                            d0 &= rem_mask0; d1 &= rem_mask1; d2 &= rem_mask2;
                        }
                        /* bytes 0 .. 11 of the group: positions 0 1 2 0 | 1 2 0 1 | 2 0 1 2 */
                        s0 += lut.opaque_lin (d0 & 0xffu) + lut.opaque_lin (d0 >> 24) + lut.opaque_lin ((d1 >> 16) & 0xffu) + lut.opaque_lin ((d2 >> 8) & 0xffu);
                        s1 += lut.opaque_lin ((d0 >> 8) & 0xffu) + lut.opaque_lin (d1 & 0xffu) + lut.opaque_lin (d1 >> 24) + lut.opaque_lin ((d2 >> 16) & 0xffu);
                        s2 += lut.opaque_lin ((d0 >> 16) & 0xffu) + lut.opaque_lin ((d1 >> 8) & 0xffu) + lut.opaque_lin (d2 & 0xffu) + lut.opaque_lin (d2 >> 24);
                    }
                }
                auto at_pos = [&] (uint32_t pos) { return pos == 0 ? s0 : pos == 1 ? s1 : s2; };       /* (pos is wave-uniform) */
                acc [0] += at_pos (J.pos_in [0]);
                acc [1] += at_pos (J.pos_in [1]);
                acc [2] += at_pos (J.pos_in [2]);
                acc [3] += 255u * n_int;
            }
            else
            {
            /* interior pixels four at a time, converted in one branch-free block (`un': the conversion's flavour, fixed
             * per wave); a pixel past the lane's share is a zero word with its channels masked off. The last group holds
             * the 1-3 pixels the longest column has left (wave-uniform: a job constant) -- a box of 9.14 pixels has 8 or 9
             * interior ones, and a third group of four converted 12 pixels for them (4096^2 -> 448^2 gamma-correct:
             * 14 conversions per column and row where 11 do; profiles/r04_boxh_tail_ab.txt) */
            auto group = [&] (auto n_tag, const uint32_t k)
            {
                constexpr uint32_t N = decltype (n_tag)::value;
                uint32_t w4 [N], c4 [N][4];
                /* (a pixel past the lane's share is the neighbouring column's: staged bytes, converted for nothing and
                 * masked off below -- cheaper than a select per pixel in front of the read) */
#pragma unroll
                for (uint32_t u = 0; u < N; u++)
                    w4 [u] = box2_px_word<AL> (J, rowb, off + u * bpp);
                constexpr int UK = std::remove_reference_t<decltype (un)>::KIND;
                bool opaque = false;
                if constexpr (UK == WU_LIN_P || UK == WU_LIN_U)
                if (try_opaque)
                {
                    /* photographs with an alpha channel: a group that is opaque on the whole wave takes the fused table.
                     * The first translucent group ends the looking for the rest of this wave's rows: on translucent
                     * images the test itself costs 9 % (profiles/r03j_box_opaque_check_ab.txt), on opaque ones the fused
                     * table saves 14 % */
                    uint32_t all = 0xffffffffu;
#pragma unroll
                    for (uint32_t u = 0; u < N; u++)
                        all &= (k + u < n_int) ? w4 [u] : 0xffffffffu;
                    opaque = wide_batch_opaque (J, all);
                    try_opaque = opaque;
                }
                if (opaque)
                {
                    const WideUnpack<WU_LIN_O> uno (J, lut);
#pragma unroll
                    for (uint32_t u = 0; u < N; u++)
                        uno (w4 [u], c4 [u]);
                }
                else
                {
#pragma unroll
                    for (uint32_t u = 0; u < N; u++)
                        un (w4 [u], c4 [u]);
                }
                if (k + N <= n_int)
                {
                    /* a whole group of the lane's share: no masks */
#pragma unroll
                    for (uint32_t u = 0; u < N; u++)
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            acc [i] += c4 [u][i];
                }
                else
                {
#pragma unroll
                    for (uint32_t u = 0; u < N; u++)
                    {
                        const uint32_t m = (k + u < n_int) ? 0xffffffffu : 0u;
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            acc [i] += c4 [u][i] & m;
                    }
                }
                off += N * bpp;
            };
            const uint32_t nfull = J.box_nmax & ~3u;
            uint32_t k = 0;
            for (; k < nfull; k += 4)
                group (std::integral_constant<uint32_t, 4> (), k);
            switch (J.box_nmax & 3u)                /* wave-uniform */
            {
                case 1: group (std::integral_constant<uint32_t, 1> (), k); break;
                case 2: group (std::integral_constant<uint32_t, 2> (), k); break;
                case 3: group (std::integral_constant<uint32_t, 3> (), k); break;
                default: break;
            }
            }
            if (LPC > 1)
            {
#pragma unroll
                for (uint32_t m = 1; m < (uint32_t) LPC; m <<= 1)
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        acc [i] += smol_shfl_xor (acc [i], m);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
                acc [i] = box_scale<true> (acc [i], J.span_mul_x);
            res = u32x2a8 { acc [0] | (acc [1] << 16), acc [2] | (acc [3] << 16) };
        }
        else
        {
            /* unassociated input is premultiplied per pixel (smolscale-generic.c:238-244, 415-464) */
            const bool pm = J.unassoc_in;
            const uint32_t ain = J.pos_in [3];
            uint32_t w;
            Px2 a = { 0, 0 };
            if (edges)
            {
                w = box2_px_word<AL> (J, rowb, off);
                a = px2_mulw (pm ? premul_px2 (w, ain) : px2_from_word (w), wl);
                w = box2_px_word<AL> (J, rowb, last_ofs + mis);
                a = px2_add (a, px2_mulw (pm ? premul_px2 (w, ain) : px2_from_word (w), fr));
            }
            off = int_ofs + mis;
            if (bpp == 3)
            {
                /* 24-bit pixels in groups of 4 = 12 bytes = 3 dwords (b0 b1 b2 b0 | b1 b2 b0 b1 |
                 * b2 b0 b1 b2): each channel's four bytes are picked out and summed by three
                 * v_dot4_u32_u8 with one-hot byte weights -- 9 per group instead of 9 unpacking
                 * operations + 6 packed adds. The 0-3 pixels left over are one more group with
                 * the bytes past the span masked off (masks fixed per lane, set up per tile). */
                constexpr uint32_t W03 = 0x01000001u, W1 = 0x00000100u, W2 = 0x00010000u;
                const uint32_t ngrp = n_int >> 2;
                uint32_t c0 = 0, c1 = 0, c2 = 0;
                const uint32_t sh = off & 3u;
                const uint32_t *pp = reinterpret_cast<const uint32_t *> (rowb + (off & ~3u));
                for (uint32_t g = 0; g <= J.box_nmax; g += 4, pp += 3)
                {
                    if ((g >> 2) <= ngrp)
                    {
                        const uint32_t q0 = pp [0], q1 = pp [1], q2 = pp [2], q3 = pp [3];
                        uint32_t d0 = __builtin_amdgcn_alignbyte (q1, q0, sh);
                        uint32_t d1 = __builtin_amdgcn_alignbyte (q2, q1, sh);
                        uint32_t d2 = __builtin_amdgcn_alignbyte (q3, q2, sh);
                        if ((g >> 2) == ngrp)
                        {
                            d0 &= rem_mask0; d1 &= rem_mask1; d2 &= rem_mask2;
                        }
                        c0 = smol_udot4 (d0, W03, c0); c0 = smol_udot4 (d1, W2, c0); c0 = smol_udot4 (d2, W1, c0);
                        c1 = smol_udot4 (d0, W1, c1); c1 = smol_udot4 (d1, W03, c1); c1 = smol_udot4 (d2, W2, c1);
                        c2 = smol_udot4 (d0, W2, c2); c2 = smol_udot4 (d1, W1, c2); c2 = smol_udot4 (d2, W03, c2);
                    }
                }
                Px2 grp;
                grp.lo = c0 | (c2 << 16);
                grp.hi = c1 | ((n_int * 255u) << 16);                     /* alpha = 0xff per pixel */
                a = px2_add (a, grp);
            }
            else
            {
                /* interior pixels four at a time (a zero word adds nothing in this class), the four of a group in a per-lane
                 * rotated order: at a ratio of 16 or 32 the lanes' spans start 64 or 128 bytes apart and every read of the
                 * plain order lands in 4 or 2 banks (4096 -> 256: 130 us against 108-111 at ratios 14 and 15; as k_tile's
                 * box_rot) */
                const uint32_t rot = lane & 3u;
                for (uint32_t k = 0; k < J.box_nmax; k += 4)
                {
                    uint32_t w4 [4];
#pragma unroll
                    for (uint32_t u0 = 0; u0 < 4; u0++)
                    {
                        const uint32_t u = (u0 + rot) & 3u;
                        w4 [u0] = (k + u < n_int) ? box2_px_word<AL> (J, rowb, off + u * bpp) : 0u;
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++)
                        a = px2_add (a, pm ? premul_px2 (w4 [u], ain) : px2_from_word (w4 [u]));
                    off += 4u * bpp;
                }
            }
            if (LPC > 1)
            {
#pragma unroll
                for (uint32_t m = 1; m < (uint32_t) LPC; m <<= 1)
                {
                    Px2 o2;
                    o2.lo = smol_shfl_xor (a.lo, m);
                    o2.hi = smol_shfl_xor (a.hi, m);
                    a = px2_add (a, o2);
                }
            }
            a = px2_box_scale (a, J.span_mul_x);
            res = u32x2a8 { a.lo, a.hi };
        }

        if (x < xn && sub == 0)
            *reinterpret_cast<SMOL_AS_GLOBAL u32x2a8 *> (gptr (J.tmp) + ((size_t) (r - J.tmp_row0) * J.w_out + x) * 8) = res;
    }
    };
    if constexpr (!WIDE)
    {
        const WideUnpack<WU_ANY> un (J, lut);                       /* (unused by the 64bpp class) */
        if (J.in_dword_ok)
            rows (BoxAligned<true> (), un);
        else
            rows (BoxAligned<false> (), un);
    }
    else
    {
    /* 128bpp class: the conversion's flavour is fixed here, once per wave (smol_kernels.hip.h WideUnpack) */
    int uk = wide_unpack_kind (J);
    if (uk == WU_ANY && !J.p16 && !J.linear && !J.unassoc_in && J.bpp_in == 4)
        uk = WU_FLAT_P;
    auto go = [&] (const auto &un)
    {
        if (J.in_dword_ok)
            rows (BoxAligned<true> (), un);
        else
            rows (BoxAligned<false> (), un);
    };
    switch (uk)
    {
        case WU_LIN_P:  go (WideUnpack<WU_LIN_P> (J, lut)); break;
        case WU_LIN_U:  go (WideUnpack<WU_LIN_U> (J, lut)); break;
        case WU_LIN_O:  go (WideUnpack<WU_LIN_O> (J, lut)); break;
        case WU_FLAT_P: go (WideUnpack<WU_FLAT_P> (J, lut)); break;
        default:        go (WideUnpack<WU_ANY> (J, lut)); break;
    }
    }
}

/* k_boxh for columns whose span does not fit the LDS (round 6): the whole wave on ONE column, its pixels read STRAIGHT from
 * memory. A box that is summed reads every pixel once, so the LDS row buffer of boxh_body buys nothing but coalesced 16-byte
 * loads -- and four waves' buffers hold 9 600 pixels per column at most: a 65 535-pixel row boxed into ONE pixel (and
 * every horizontal ratio beyond 9 600) used to end in k_general<true>, one LANE walking the row (3.3 % of
 * tools/fuzz/gpu_fuzz.py's cases, at a few per cent of the roofline). Here lane l takes the interior pixels l, l + 64,
 * l + 128 ... of the column -- one dword per lane, a wave instruction reads 256 (192 for 24-bit rows) contiguous bytes,
 * eight of them in flight --, converts (the wave's flavour, as boxh_body), sums in 32 bits; the lanes meet in a shuffle
 * butterfly (sums of exact integers: smolscale-generic.c:1481-1556 adds them in pixel order, modulo 2^32 like these),
 * lane 0 adds the two weighted edge pixels, normalises and writes the intermediate pixel. 128bpp class only (a ratio
 * beyond 255). JobDev::span_chunks == 0 marks the job (smol_jobdev.h: box2_setup, twopass_setup). */
template <typename UN>
__device__ __forceinline__ void boxh_direct_rows (const JobDev &J, const UN &un, const uint32_t lane, const uint32_t x,
                                                  const uint32_t r0, const uint32_t r1)
{
    uint32_t o0, o1, fr, wl;
    box_span (J.step_x, J.w_in, x, o0, o1, fr, wl);
    const uint32_t n_col = o1 - o0 - 1;                             /* interior pixels of the column */
    const uint32_t bpp = J.bpp_in;
    const uint32_t last_px = J.w_in - 1u;
    constexpr uint32_t U = 8;

    /* pixel p of the row as its bytes in memory order; a 24-bit pixel is the dword at its own byte address, the row's last
     * one read a byte early and shifted (nothing past the row is touched: k_lind's rule) */
    auto load_px = [&] (const gbyte *row, uint32_t p) -> uint32_t
    {
        uint32_t w;
        if (bpp == 4)
        {
#ifdef SMOL_KERNEL_EMU
            __builtin_memcpy (&w, row + (size_t) p * 4u, 4);
#else
            w = *reinterpret_cast<const SMOL_AS_GLOBAL u32a1 *> (row + (size_t) p * 4u);
#endif
            return w;
        }
        const uint32_t sh = (p == last_px && p != 0u) ? 1u : 0u;
#ifdef SMOL_KERNEL_EMU
        w = 0;
        __builtin_memcpy (&w, row + (size_t) p * 3u - sh, (p == last_px && p == 0u) ? 3 : 4);
#else
        w = *reinterpret_cast<const SMOL_AS_GLOBAL u32a1 *> (row + (size_t) p * 3u - sh);
#endif
        return __builtin_amdgcn_alignbyte (0u, w, sh) | 0xff000000u;
    };

    for (uint32_t r = r0; r <= r1; r++)
    {
        const gbyte *row = gptr (J.src) + (size_t) r * J.src_stride;
        uint32_t acc [4] = { 0, 0, 0, 0 };
        for (uint32_t base = 0; base < n_col; base += 64u * U)     /* (wave-uniform trip count) */
        {
            const uint32_t i = base + lane;
            uint32_t w [U], c [4];
#pragma unroll
            for (uint32_t u = 0; u < U; u++)
            {
                const uint32_t k = i + 64u * u;
                w [u] = load_px (row, o0 + 1u + (k < n_col ? k : 0u));  /* (past the column: its first interior pixel again, masked below) */
            }
#pragma unroll
            for (uint32_t u = 0; u < U; u++)
            {
                un (w [u], c);
                const uint32_t m = (i + 64u * u < n_col) ? 0xffffffffu : 0u;
#pragma unroll
                for (int ch = 0; ch < 4; ch++)
                    acc [ch] += c [ch] & m;
            }
        }
#pragma unroll
        for (int ch = 0; ch < 4; ch++)
            acc [ch] = wave_sum (acc [ch]);
        if (lane == 0)
        {
            uint32_t c [4];
            un (load_px (row, o0), c);
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
                acc [ch] += ((c [ch] * wl) >> 8) & 0x00ffffffu;
            un (load_px (row, o1), c);
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
                acc [ch] += ((c [ch] * fr) >> 8) & 0x00ffffffu;
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
                acc [ch] = box_scale<true> (acc [ch], J.span_mul_x);
            *reinterpret_cast<SMOL_AS_GLOBAL u32x2a8 *> (gptr (J.tmp) + ((size_t) (r - J.tmp_row0) * J.w_out + x) * 8)
                = u32x2a8 { acc [0] | (acc [1] << 16), acc [2] | (acc [3] << 16) };
        }
    }
}

__device__ __forceinline__ void boxh_direct_body (const JobDev &J, const uint8_t *lut_lds, const uint32_t lane,
                                                  const uint32_t tx, const uint32_t tr)
{
    constexpr uint32_t BOXH_ROWS = SMOL_BOXH_ROWS_WIDE;             /* source rows per wave (smol_jobdev.h: boxh_rows) */
    const uint32_t r0 = J.tmp_row0 + tr * BOXH_ROWS;
    if (r0 > J.tmp_row1 || tx >= J.w_out)
        return;
    const uint32_t r1 = (r0 + BOXH_ROWS - 1 < J.tmp_row1) ? r0 + BOXH_ROWS - 1 : J.tmp_row1;
    const LutLds lut (lut_lds, LUT_AT_LDS_0);
    /* the conversion's flavour, fixed once per wave (as boxh_body) */
    int uk = wide_unpack_kind (J);
    if (uk == WU_ANY && !J.p16 && !J.linear && !J.unassoc_in && J.bpp_in == 4)
        uk = WU_FLAT_P;
    switch (uk)
    {
        case WU_LIN_P:  boxh_direct_rows (J, WideUnpack<WU_LIN_P> (J, lut), lane, tx, r0, r1); break;
        case WU_LIN_U:  boxh_direct_rows (J, WideUnpack<WU_LIN_U> (J, lut), lane, tx, r0, r1); break;
        case WU_LIN_O:  boxh_direct_rows (J, WideUnpack<WU_LIN_O> (J, lut), lane, tx, r0, r1); break;
        case WU_FLAT_P: boxh_direct_rows (J, WideUnpack<WU_FLAT_P> (J, lut), lane, tx, r0, r1); break;
        default:        boxh_direct_rows (J, WideUnpack<WU_ANY> (J, lut), lane, tx, r0, r1); break;
    }
}

/* LPC = lanes per output pixel: 1, or 8 for tall boxes (the lanes take every 8th row of the
 * span and meet through wave shuffles) */
template <bool WIDE, int LPC>
__device__ __forceinline__ void boxv_body (const JobDev &J, const uint8_t *lut_lds,
                                           const uint32_t lane, const uint32_t tile)
{
    constexpr uint32_t COLS = TILE_COLS / LPC;
    const uint32_t sub = lane % LPC;
    const uint32_t tiles_x = (J.w_out + COLS - 1) / COLS;
    const uint32_t tx = tile % tiles_x;
    const uint32_t row = tile / tiles_x;                            /* output row of the strip */

    if (row >= J.n_rows)
        return;

    const uint32_t x = tx * COLS + lane / LPC;
    const bool active = x < J.w_out;
    if (LPC == 1 && !active)
        return;
    const uint32_t xc = active ? x : J.w_out - 1;                   /* idle lane groups shadow the last column (shuffles below) */

    const LutLds lut (lut_lds, LUT_AT_LDS_0);       /* (boxh_entry / boxv_entry: lut_expect_lds_0) */
    uint32_t y0, y1, w2, w1;
    box_span (J.step_y, J.h_in, J.first_row + row, y0, y1, w2, w1);
    /* 128bpp-class rows weight the last row by w2 - 1 and skip it when w2 == 0
     * (smolscale-generic.c:2236-2253); 64bpp-class rows use w2 as is (:2042-2069) */
    const uint32_t last_w = WIDE ? (w2 ? w2 - 1 : 0) : w2;
    const uint32_t y_end = (WIDE && w2 == 0) ? y1 - 1 : y1;
    const size_t ipitch = (size_t) J.w_out * 8;
    const gbyte *ip = gptr (J.tmp) + ((size_t) (y0 + sub - J.tmp_row0) * J.w_out + xc) * 8;
    uint8_t *out_row = J.dst + (size_t) row * J.dst_stride;

    if (WIDE)
    {
        uint32_t acc [4] = { 0, 0, 0, 0 };
        if (LPC == 1 && dispatch::twopass_hbox (J))
        {
            /* pass 1 kept the box sums in its registers (JobDev::box2 == 3): what is left is adding the splits' partial sums */
            const uint32_t S = dispatch::hbox_splits (J);
            const gbyte *pp = gptr (J.tmp) + ((size_t) row * S * J.w_out + xc) * 16u;
            for (uint32_t sp = 0; sp < S; sp++, pp += (size_t) J.w_out * 16u)
            {
                const u32x4 t = *reinterpret_cast<const SMOL_AS_GLOBAL u32x4 *> (pp);
                acc [0] += t.x; acc [1] += t.y; acc [2] += t.z; acc [3] += t.w;
            }
        }
        else
        /* rows eight at a time: the loads of a group are in flight together */
        for (uint32_t r = y0 + sub; r <= y_end; r += 8 * LPC, ip += 8 * LPC * ipitch)
        {
            u32x2a8 v8 [8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++)
                v8 [u] = (r + u * LPC <= y_end) ? *reinterpret_cast<const SMOL_AS_GLOBAL u32x2a8 *> (ip + u * LPC * ipitch) : u32x2a8 { 0, 0 };
#pragma unroll
            for (uint32_t u = 0; u < 8; u++)
            {
                const uint32_t c [4] = { v8 [u].x & 0xffffu, v8 [u].x >> 16, v8 [u].y & 0xffffu, v8 [u].y >> 16 };
                const uint32_t rr = r + u * LPC;
                const uint32_t w = (rr == y0) ? w1 : ((rr == y1) ? last_w : 0x10000u);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc [i] += (w == 0x10000u) ? c [i] : (((c [i] * w) >> 8) & 0x00ffffffu);
            }
        }
        if (LPC > 1)
        {
#pragma unroll
            for (uint32_t m = 1; m < (uint32_t) LPC; m <<= 1)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc [i] += smol_shfl_xor (acc [i], m);
        }
        uint32_t o [4];
#pragma unroll
        for (int i = 0; i < 4; i++)
            o [i] = box_scale<true> (acc [i], J.span_mul_y);
        if (active && sub == 0)
            store_px<true> (J, out_row, x, o, lut);
    }
    else
    {
        Px2 acc = { 0, 0 };
        for (uint32_t r = y0 + sub; r <= y_end; r += 8 * LPC, ip += 8 * LPC * ipitch)
        {
            u32x2a8 v8 [8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++)
                v8 [u] = (r + u * LPC <= y_end) ? *reinterpret_cast<const SMOL_AS_GLOBAL u32x2a8 *> (ip + u * LPC * ipitch) : u32x2a8 { 0, 0 };
#pragma unroll
            for (uint32_t u = 0; u < 8; u++)
            {
                const Px2 p = { v8 [u].x, v8 [u].y };
                const uint32_t rr = r + u * LPC;
                if (rr == y0)
                    acc = px2_add (acc, px2_mulw (p, w1));
                else if (rr == y1)
                    acc = px2_add (acc, px2_mulw (p, last_w));
                else
                    acc = px2_add (acc, p);
            }
        }
        if (LPC > 1)
        {
#pragma unroll
            for (uint32_t m = 1; m < (uint32_t) LPC; m <<= 1)
            {
                Px2 o2;
                o2.lo = smol_shfl_xor (acc.lo, m);
                o2.hi = smol_shfl_xor (acc.hi, m);
                acc = px2_add (acc, o2);
            }
        }
        if (active && sub == 0)
            tile_store<OUT_ANY> (J, out_row, x, px2_box_scale (acc, J.span_mul_y));
    }
}

/* 4 waves per workgroup, each on its own tile; the 128bpp class shares one LDS copy of the tables */
/* LPC64: the kernel of the jobs with 16 or 64 lanes per column (a kernel of its own: with a third body in it the common
 * kernel ran 3-4 % slower on the rows that never take it) */
template <bool WIDE, bool LPC64 = false>
__device__ __forceinline__ void boxh_entry (const JobDev &J, uint8_t *lds)
{
#ifdef SMOL_KERNEL_EMU
    static uint8_t lut_lds [LUT_LDS_BYTES];
    if (WIDE)
        lut_fill_lds (lut_lds, 0, 1);
    /* host emulation: one wave per block, blocks numbered row group by row group */
    const uint32_t e_tiles_x = (J.w_out + 64u / J.tile_lpc - 1) / (64u / J.tile_lpc);
    if (LPC64)
    {
        if (J.tile_lpc == 16)
            boxh_body<WIDE, LPC64 ? 16 : 8> (J, lds, lut_lds, threadIdx.x, blockIdx.x % e_tiles_x, blockIdx.x / e_tiles_x);
        else if (WIDE && J.span_chunks == 0)
            boxh_direct_body (J, lut_lds, threadIdx.x, blockIdx.x % e_tiles_x, blockIdx.x / e_tiles_x);
        else
            boxh_body<WIDE, LPC64 ? 64 : 8> (J, lds, lut_lds, threadIdx.x, blockIdx.x % e_tiles_x, blockIdx.x / e_tiles_x);
    }
    else if (J.tile_lpc == 8)
        boxh_body<WIDE, 8> (J, lds, lut_lds, threadIdx.x, blockIdx.x % e_tiles_x, blockIdx.x / e_tiles_x);
    else
        boxh_body<WIDE, 1> (J, lds, lut_lds, threadIdx.x, blockIdx.x % e_tiles_x, blockIdx.x / e_tiles_x);
#else
    const uint32_t wave = smol_wave_index ();       /* a SCALAR: everything derived from the tile index stays wave-uniform */
    /* the workgroup's 4 waves: 4 column tiles of one row group, or 4 row groups of one column tile (box2_setup) */
    const uint32_t tx = J.box2_ygroup ? blockIdx.x : blockIdx.x * 4u + wave;
    const uint32_t tr = J.box2_ygroup ? blockIdx.y * 4u + wave : blockIdx.y;
    if (WIDE)
    {
        lut_expect_lds_0 (lds);
        lut_fill_lds (lds, threadIdx.x, 256);
        __syncthreads ();
    }
    const uint32_t lut_bytes = WIDE ? ((LUT_LDS_BYTES + 15u) & ~15u) : 0u;
    if (LPC64)
    {
        uint8_t *buf = lds + lut_bytes + wave * J.span_chunks * 16u;
        if (J.tile_lpc == 16)
            boxh_body<WIDE, LPC64 ? 16 : 8> (J, buf, lds, threadIdx.x & 63u, tx, tr);
        else if (WIDE && J.span_chunks == 0)
            boxh_direct_body (J, lds, threadIdx.x & 63u, tx, tr);        /* (a span too long for the LDS: straight from memory) */
        else
            boxh_body<WIDE, LPC64 ? 64 : 8> (J, buf, lds, threadIdx.x & 63u, tx, tr);
    }
    else if (J.tile_lpc == 8)
        boxh_body<WIDE, 8> (J, lds + lut_bytes + wave * J.span_chunks * 16u, lds, threadIdx.x & 63u, tx, tr);
    else
        boxh_body<WIDE, 1> (J, lds + lut_bytes + wave * J.span_chunks * 16u, lds, threadIdx.x & 63u, tx, tr);
#endif
}

template <bool WIDE>
__device__ __forceinline__ void boxv_entry (const JobDev &J, uint8_t *lds)
{
#ifdef SMOL_KERNEL_EMU
    static uint8_t lut_lds [LUT_LDS_BYTES];
    if (WIDE)
        lut_fill_lds (lut_lds, 0, 1);
    if (J.box2_vlpc == 8)
        boxv_body<WIDE, 8> (J, lut_lds, threadIdx.x, blockIdx.x);
    else
        boxv_body<WIDE, 1> (J, lut_lds, threadIdx.x, blockIdx.x);
#else
    if (WIDE)
    {
        lut_expect_lds_0 (lds);
        lut_fill_lds (lds, threadIdx.x, 256);
        __syncthreads ();
    }
    if (J.box2_vlpc == 8)
        boxv_body<WIDE, 8> (J, lds, threadIdx.x & 63u, blockIdx.x * 4u + smol_wave_index ());
    else
        boxv_body<WIDE, 1> (J, lds, threadIdx.x & 63u, blockIdx.x * 4u + smol_wave_index ());
#endif
}

#ifdef SMOL_KERNEL_EMU
__global__ void k_boxh_one (const JobDev J)
{
    if (twopass_lin_h (J)) { linh_dispatch (J, smol_dyn_lds); return; }
    if (J.wide && J.tile_lpc >= 16) boxh_entry<true, true> (J, smol_dyn_lds);
    else if (J.wide) boxh_entry<true> (J, smol_dyn_lds); else boxh_entry<false> (J, smol_dyn_lds);
}

__global__ void k_boxv_one (const JobDev J)
{
    if (twopass_lin_v (J)) { linv_dispatch (J, smol_dyn_lds); return; }
    if (J.wide) boxv_entry<true> (J, smol_dyn_lds); else boxv_entry<false> (J, smol_dyn_lds);
}
#else
template <bool WIDE, bool LPC64 = false>
__global__ void __launch_bounds__ (256)
k_boxh_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.z);
    boxh_entry<WIDE, LPC64> (J, smol_dyn_lds);
}

template <bool WIDE, bool LPC64 = false>
__global__ void __launch_bounds__ (256)
k_boxh_one (const JobDev J)
{
    boxh_entry<WIDE, LPC64> (J, smol_dyn_lds);
}

template <bool WIDE>
__global__ void __launch_bounds__ (256)
k_boxv_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.y);
    boxv_entry<WIDE> (J, smol_dyn_lds);
}

template <bool WIDE>
__global__ void __launch_bounds__ (256)
k_boxv_one (const JobDev J)
{
    boxv_entry<WIDE> (J, smol_dyn_lds);
}
#endif

} /* namespace smolhip */
