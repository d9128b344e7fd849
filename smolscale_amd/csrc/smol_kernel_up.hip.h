/* smol_kernel_up.hip.h -- k_up: bilinear MAGNIFICATION on both axes, 64bpp class
 * (BASELINE config 3 and every other upscale). Included at the end of
 * smol_kernels.hip.h, after smol_kernel_tile.hip.h whose packed-lane helpers and
 * pixel stores it shares.
 *
 * Magnification is write-bound: one source pixel feeds many outputs, so the
 * kernel is organised around the stores. One wave owns a tile of 256 output
 * columns x tile_rows output rows; a lane owns 4 adjacent columns (one 16-byte
 * store per output row).
 *
 *   stage  With both steps below 1.0 a tile of up to 64 output rows touches at
 *          most UP_NSRC source rows (the host sizes tile_rows for that) and a short
 *          span of each. Every source pixel of the tile is loaded ONCE, premultiplied
 *          if the input is unassociated (smolscale-generic.c:238-244, 415-464),
 *          split into packed 16-bit lanes and parked in LDS (8 bytes per pixel).
 *   H      a lane reads the two pixels under each of its 4 columns for every source
 *          row and keeps the horizontal result (:1290-1361) as one pixel word per
 *          (row, column) in registers.
 *   V      output rows then leave in order: row r blends the register rows oy(r)
 *          and oy(r)+1 with weight F(r) (:1648-1719) -- two packed multiply-adds and
 *          one v_perm per pixel -- and is stored at once. oy/F come from a schedule
 *          the lanes computed once (lane r holds row r's entry), fetched with
 *          v_readlane. One wave per workgroup: no barriers, and nothing ever waits
 *          for a store.
 *
 * Tap arithmetic is the packed 16-bit form of k_tile (tap2): for 8-bit values
 * q + (((p - q) * F) >> 8) == (p * F + q * (256 - F)) >> 8. */

#pragma once

#include <type_traits>

namespace smolhip {

constexpr int UP_CPL = 4;           /* output columns per lane                   */
constexpr int UP_NSRC = 10;         /* source rows a tile may touch (host: tile_setup_up) */

enum : int
{
    UP_IN_A4 = 0,   /* 32-bit pixels at 4-byte aligned addresses */
    UP_IN_ANY = 1   /* 24-bit pixels, or 32-bit ones at odd addresses */
};

/* one source pixel as a pixel word (24-bit input: alpha = 0xff) */
template <int INK>
__device__ __forceinline__ uint32_t up_load_px (const JobDev &J, const gbyte *rowp, uint32_t x)
{
    if (INK == UP_IN_A4)
        return *reinterpret_cast<const SMOL_AS_GLOBAL uint32_t *> (rowp + (size_t) x * 4);

    /* aligned dwords around the pixel; a dword holding at least one wanted byte
     * never leaves the buffer's last page, one holding none is not touched */
    const uint32_t nb = J.bpp_in;
    const uintptr_t a = reinterpret_cast<uintptr_t> (rowp) + (size_t) x * nb;
    const uint32_t sh = (uint32_t) a & 3u;
    const SMOL_AS_GLOBAL uint32_t *d = reinterpret_cast<const SMOL_AS_GLOBAL uint32_t *> (a - sh);
    const uint32_t d0 = d [0];
    uint32_t d1 = 0;
    if (sh + nb > 4u)
        d1 = d [1];
    const uint32_t w = __builtin_amdgcn_alignbyte (d1, d0, sh);
    return (nb == 3) ? (w | 0xff000000u) : w;
}

constexpr int UP_STAGE_UNROLL = 8;
constexpr uint32_t UP_SEL_HIBYTES = 0x07030501u;    /* v_perm: high byte of each 16-bit lane of {S0 = hi, S1 = lo} */

typedef uint32_t u32x4a8 __attribute__ ((ext_vector_type (4), aligned (8)));
typedef uint32_t u32x2a8 __attribute__ ((ext_vector_type (2), aligned (8)));

/* d * F + b per 16-bit lane (modulo 2^16; exact here, see up_body) */
__device__ __forceinline__ uint32_t up_mad (uint32_t d, uint32_t F2, uint32_t b)
{
    return as_u32 (as_u16x2 (d) * as_u16x2 (F2) + as_u16x2 (b));
}

/* the lane's four 32-bit pixels as one 16-byte store, the wave's kilobyte back to back (k_up, its premul16 flavour, k_upg).
 * Rows that start on 64-byte boundaries: streaming. Rows that start anywhere in a 128-byte line: the lanes whose 16 bytes
 * touch the run's first or last line store write-back, so that the two parts of a line -- the other comes from the
 * neighbouring tile at another time -- meet in the L2, the whole lines in between keep streaming (the comment in up_body). */
__device__ __forceinline__ void up_store16 (const JobDev &J, gbyte *out_row, uint32_t xl, uint32_t lane, u32x4 v)
{
    if (!J.out_wt)
    {
        const uint32_t a = (uint32_t) (uintptr_t) out_row + xl * 4u;
        const uint32_t run = a - lane * 16u;
        const uint32_t lo_b = (run + 127u) & ~127u, hi_b = (run + 1024u) & ~127u;
        if (a < lo_b || a + 16u > hi_b)
        {
#ifdef SMOL_KERNEL_EMU
            __builtin_memcpy (out_row + (size_t) xl * 4, &v, 16);
#else
            *reinterpret_cast<SMOL_AS_GLOBAL u32x4a4 *> (out_row + (size_t) xl * 4) = v;
#endif
        }
        else
            store_out128_nt (out_row + (size_t) xl * 4, v);
    }
    else
        store_nt128 (out_row + (size_t) xl * 4, v);
}

template <int INK, int OUT>
__device__ __forceinline__ void up_body (const JobDev &J, uint8_t *lds)
{
    constexpr uint32_t TW = TILE_COLS * UP_CPL;

    const uint32_t lane = threadIdx.x;
    const uint32_t tiles_x = (J.w_out + TW - 1) / TW;
    /* each XCD (blocks go round-robin over the 8 of them) takes a contiguous band of
     * tiles, so horizontally adjacent tiles are written from the same L2 at about
     * the same time; the host rounds the grid up to a multiple of 8 */
    const uint32_t bid = J.tile_band ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t tx = bid % tiles_x;
    const uint32_t ty = bid / tiles_x;
    const uint32_t yrel0 = ty * J.tile_rows;

    if (yrel0 >= J.n_rows)
        return;

    const uint32_t nrows = (yrel0 + J.tile_rows < J.n_rows) ? J.tile_rows : J.n_rows - yrel0;   /* <= 64 */
    const uint32_t x0 = tx * TW;
    const uint32_t xn = (x0 + TW < J.w_out) ? x0 + TW : J.w_out;
    const uint32_t xl = x0 + lane * UP_CPL;

    /* ---- horizontal taps of this lane's columns ------------------------------- */
    uint32_t tap_ofs [UP_CPL], tap_f2 [UP_CPL];
#pragma unroll
    for (int c = 0; c < UP_CPL; c++)
    {
        const uint32_t xc = (xl + c < xn) ? xl + c : xn - 1;
        uint32_t f;
        bilin_tap (J.frac0_x + (uint64_t) xc * J.step_x, J.w_in, tap_ofs [c], f);
        tap_f2 [c] = f | (f << 16);
    }
    const uint32_t sx0 = smol_readlane (tap_ofs [0], 0);       /* first source pixel of the tile */

    /* ---- vertical schedule: lane r holds (source row, weight) of tile row r ---- */
    uint32_t my_oy, my_f;
    {
        const uint32_t r = (lane < nrows) ? lane : nrows - 1;
        bilin_tap (J.frac0_y + (uint64_t) (J.first_row + yrel0 + r) * J.step_y, J.h_in, my_oy, my_f);
    }
    const uint32_t oy_first = smol_readlane (my_oy, 0);
    const uint32_t oy_last = smol_readlane (my_oy, nrows - 1);
    const uint32_t sched = (my_oy - oy_first) | (my_f << 16);
    uint32_t n_src = oy_last - oy_first + 2;
    if (n_src > (uint32_t) UP_NSRC)
        n_src = UP_NSRC;                                /* cannot happen: up_setup sizes tile_rows */

    /* ---- stage: every source pixel of the tile, once ------------------------------
     * span_px pixels (host bound, >= what the taps reach) of each of the n_src rows:
     * loaded, premultiplied if the input is unassociated, split into the two packed
     * 16-bit-lane halves (bytes 0,2 | bytes 1,3) and written to LDS, 8 bytes per
     * pixel; pixel k of row s sits at 8 * (s * span_px + k). */
    {
        const gbyte *srow0 = gptr (J.src) + (size_t) oy_first * J.src_stride;
        const uint32_t ain = (J.bpp_in == 4) ? J.pos_in [3] : 3u;
        const bool premul_in = J.unassoc_in;
        const uint32_t span = J.span_px;
        const uint32_t total = n_src * span;

        for (uint32_t i0 = 0; i0 < total; i0 += 64u * UP_STAGE_UNROLL)
        {
            uint32_t w [UP_STAGE_UNROLL];
#pragma unroll
            for (int u = 0; u < UP_STAGE_UNROLL; u++)
            {
                const uint32_t i = i0 + (uint32_t) u * 64u + lane;
                const uint32_t srow = (uint32_t) (((uint64_t) i * J.span_chunks_inv) >> 32);   /* i / span */
                const uint32_t k = i - srow * span;
                uint32_t x = sx0 + k;
                if (x > J.w_in - 1)
                    x = J.w_in - 1;
                w [u] = 0;
                if (i < total)
                    w [u] = up_load_px<INK> (J, srow0 + (size_t) srow * J.src_stride, x);
            }
#pragma unroll
            for (int u = 0; u < UP_STAGE_UNROLL; u++)
            {
                const uint32_t i = i0 + (uint32_t) u * 64u + lane;
                const Px2 v = premul_in ? premul_px2 (w [u], ain) : px2_from_word (w [u]);
                const u32x2a8 pair = { v.lo, v.hi };
                if (i < total)
                    *reinterpret_cast<u32x2a8 *> (lds + (size_t) i * 8) = pair;
            }
        }
    }
    /* unassociated output (round 5): the inverse-alpha table behind the staged rows -- a look-up in global memory makes the
     * compiler drain every store in flight in front of each row (k_tile's finding, smol_kernel_tile.hip.h tile_pack) */
    uint32_t *inv_lds = reinterpret_cast<uint32_t *> (lds + ((UP_NSRC * J.span_px * 8u + 15u) & ~15u));
    if (OUT == OUT_U4)
    {
#pragma unroll
        for (int i = 0; i < 4; i++)
            inv_lds [i * 64 + lane] = smol_inv_p8_lut [i * 64 + lane];
    }
    smol_wave_sync ();                                  /* one wave: LDS operations complete in order */

    /* ---- H: horizontal result of every (source row, column), one word each -------
     * (smolscale-generic.c:1290-1361). p * F + q * (256 - F) is at most 255 * 256,
     * so the high byte of each 16-bit lane is the result and one v_perm gathers
     * the four of them back into a pixel word. */
    uint32_t hw [UP_NSRC][UP_CPL];
#pragma unroll
    for (int s = 0; s < UP_NSRC; s++)
    {
#pragma unroll
        for (int c = 0; c < UP_CPL; c++)
            hw [s][c] = 0;
        if ((uint32_t) s < n_src)                       /* wave-uniform */
        {
#pragma unroll
            for (int c = 0; c < UP_CPL; c++)
            {
                const uint32_t k = (uint32_t) s * J.span_px + (tap_ofs [c] - sx0);
                const u32x4a8 pq = *reinterpret_cast<const u32x4a8 *> (lds + (size_t) k * 8);
                const uint32_t g2 = 0x01000100u - tap_f2 [c];
                const uint32_t lo = up_mad (pq.x, tap_f2 [c], as_u32 (as_u16x2 (pq.z) * as_u16x2 (g2)));
                const uint32_t hi = up_mad (pq.y, tap_f2 [c], as_u32 (as_u16x2 (pq.w) * as_u16x2 (g2)));
                hw [s][c] = __builtin_amdgcn_perm (hi, lo, UP_SEL_HIBYTES);
            }
        }
    }

    /* ---- V: output rows in order (:1648-1719) ------------------------------------
     * For the rows between source rows s and s+1:  top * F + bot * (256 - F)
     *   == (top - bot) * F + (bot << 8)   modulo 2^16 per lane, and the true value
     * fits 16 bits, so d = top - bot and b = bot << 8 are set up once per source
     * row pair and an output pixel costs two packed multiply-adds and one v_perm
     * (which also reorders the bytes for the destination type). */
    const bool interior = (xn == x0 + TW) && J.out_vec16_ok;
    const bool full24 = (xn == x0 + TW) && J.out3_ok;
    uint8_t *out_row = J.dst + (size_t) yrel0 * J.dst_stride;
    uint32_t r = 0;
    uint32_t e = smol_readlane (sched, 0);
    const uint32_t M_HI = 0xff00ff00u;

#pragma unroll
    for (int s = 0; s + 1 < UP_NSRC; s++)
    {
        if (r >= nrows || (e & 0xffffu) != (uint32_t) s)
            continue;                                   /* wave-uniform */

        uint32_t d_lo [UP_CPL], d_hi [UP_CPL], b_lo [UP_CPL], b_hi [UP_CPL];
#pragma unroll
        for (int c = 0; c < UP_CPL; c++)
        {
            const uint32_t t = hw [s][c], b = hw [s + 1][c];
            d_lo [c] = as_u32 (as_u16x2 (t & PAIR_MASK) - as_u16x2 (b & PAIR_MASK));
            d_hi [c] = as_u32 (as_u16x2 ((t >> 8) & PAIR_MASK) - as_u16x2 ((b >> 8) & PAIR_MASK));
            b_lo [c] = (b << 8) & M_HI;
            b_hi [c] = b & M_HI;
        }

        while (r < nrows && (e & 0xffffu) == (uint32_t) s)
        {
            const uint32_t F = e >> 16;
            const uint32_t F2 = F | (F << 16);
            uint32_t lo [UP_CPL], hi [UP_CPL];
#pragma unroll
            for (int c = 0; c < UP_CPL; c++)
            {
                lo [c] = up_mad (d_lo [c], F2, b_lo [c]);
                hi [c] = up_mad (d_hi [c], F2, b_hi [c]);
            }

            if (OUT == OUT_P4 && interior)
            {
                const u32x4 v = { __builtin_amdgcn_perm (hi [0], lo [0], J.up_perm), __builtin_amdgcn_perm (hi [1], lo [1], J.up_perm),
                                  __builtin_amdgcn_perm (hi [2], lo [2], J.up_perm), __builtin_amdgcn_perm (hi [3], lo [3], J.up_perm) };
                /* non-temporal, not write-through: this kernel is bound by its stores, and write-through 16-byte
                 * stores run config 3 at 114 us against 88-93 us (profiles/r02m_store_policy.txt) */
                /* Rows that start anywhere in a 128-byte line (every output width that is not a multiple of 32 pixels):
                 * the wave's kilobyte is seven whole lines and two partial ones, and a streaming store of a PART of a line
                 * is a read-modify-write in memory -- the other part comes from the neighbouring tile at another time
                 * (1920x1080 x3: 104 us on 16-byte multiples, 149 us on an odd width). The lanes whose 16 bytes touch
                 * the run's first or last line store write-back, so that the two parts meet in the L2 (neighbouring tiles
                 * run on one XCD: tile_band), the whole lines in between keep streaming: x2.4 98 -> 86 us, x2.6 116 -> 95,
                 * x3 149 -> 120, x4 268 -> 231; every lane write-back 123 / 141 / 174 / 282, write-through 147 / 164 / 206 / 390
                 * (profiles/r05_up_handover.txt). Distinct instructions -- an asm store and a plain one: the compiler
                 * merges two C stores that differ only in their cache policy. */
                up_store16 (J, gptr (out_row), xl, lane, v);
            }
            else if (OUT == OUT_U4 && interior)
            {
                /* unassociated 32-bit output: un-premultiplied in registers through the LDS copy of the table, then the
                 * 16-byte store form */
                uint32_t w [UP_CPL];
#pragma unroll
                for (int c = 0; c < UP_CPL; c++)
                {
                    Px2 o;
                    o.lo = (lo [c] >> 8) & PAIR_MASK;
                    o.hi = (hi [c] >> 8) & PAIR_MASK;
                    w [c] = tile_pack<OUT_U4> (J, o, inv_lds);
                }
                const u32x4 v = { w [0], w [1], w [2], w [3] };
                up_store16 (J, gptr (out_row), xl, lane, v);
            }
            else if (OUT == OUT_ANY && full24 && UP_CPL == 4)
            {
                /* 24-bit output (round 5): the lane's four pixels are twelve consecutive bytes -- three permutes cut the
                 * three dwords out of the four pixel words, an 8-byte and a 4-byte store at the pixels' own address (global
                 * stores need no alignment on gfx950), the wave's 768 bytes back to back. 1920 x 1080 RGBA -> RGB8 x4: k_tile's shifted
                 * dword stores 111 us, this 86; x2.6 66 -> 55; unassociated -> BGR8 x4 on an odd width 158 -> 100; RGB8 sources
                 * x2.6 67 -> 65, x4 110 -> 87, x4 odd 140 -> 100, x8 100 -> 75 (profiles/r05_up_rgb8.txt). */
                const uint32_t w0 = __builtin_amdgcn_perm (hi [0], lo [0], J.up_perm), w1 = __builtin_amdgcn_perm (hi [1], lo [1], J.up_perm);
                const uint32_t w2 = __builtin_amdgcn_perm (hi [2], lo [2], J.up_perm), w3 = __builtin_amdgcn_perm (hi [3], lo [3], J.up_perm);
                const uint32_t d0 = __builtin_amdgcn_perm (w1, w0, 0x04020100u);
                const uint32_t d1 = __builtin_amdgcn_perm (w2, w1, 0x05040201u);
                const uint32_t d2 = __builtin_amdgcn_perm (w3, w2, 0x06050402u);
                /* streaming stores: the kernel is bound by them (x4: 111 us write-back, 86 streaming; x8 92 -> 75; x3 66 ->
                 * 62; x2.6 on an odd width 62 -> 65). Tried and not kept: write-back for the lanes that touch the run's first
                 * or last 128-byte line, as the 16-byte form above does (99-103 us at x4); 16-byte stores instead -- through
                 * the wave's LDS on 48 lanes (107 against 97 write-back), or within each quad of lanes by three DPP moves
                 * (128 against 97); one unaligned dword load per 24-bit source pixel (no difference) */
                store_out96_nt (gptr (out_row) + (size_t) xl * 3, d0, d1, d2);
            }
            else
            {
                Px2 o [UP_CPL];
#pragma unroll
                for (int c = 0; c < UP_CPL; c++)
                {
                    o [c].lo = (lo [c] >> 8) & PAIR_MASK;
                    o [c].hi = (hi [c] >> 8) & PAIR_MASK;
                }
                tile_store_cols<OUT, UP_CPL, true> (J, out_row, xl, xn, interior, full24, o, OUT == OUT_U4 ? inv_lds : smol_inv_p8_lut);
            }
            out_row += J.dst_stride;
            r++;
            if (r < nrows)
                e = smol_readlane (sched, r);
        }
    }
}

/* ------------------------------------------------------------------------- *
 * k_up, premul16 flavour: unassociated -> unassociated without sRGB, the       *
 * 128bpp-class default for RGBA8 data (smolscale-generic.c:522-560 unpack,    *
 * :1036-1100 pack). Same organisation; channels are 16 bit (c * a, and        *
 * (a << 8) | 0x80 for alpha), two per register, taps are 24-bit multiplies:   *
 *      q + (((p - q) * F) >> 8)  ==  (p * F + q * (256 - F)) >> 8             *
 * (an integer identity, both sides below 2^24), and an output pixel is        *
 * un-premultiplied through the inverse table (LDS copy) as it leaves.         *
 * ------------------------------------------------------------------------- */
constexpr int OUT_W16 = 3;          /* k_up<IN, OUT_W16>: the premul16 flavour */
constexpr int UP_NSRC_W = 8;        /* source rows per tile (register budget: 2 words per row and column) */

/* (p * F + q * G) >> 8 on both 16-bit halves of packed channel pairs */
/* DOT: one v_perm + one v_dot2_u32_u16 per half instead of two extractions and two 24-bit multiplies --
 * premul16 magnification 69 -> 62 us; the table-driven flavours (k_upg) lose 2 % with it, as k_downg does
 * (profiles/r02u_down_dot2.txt), and keep the multiplies */
template <bool DOT>
__device__ __forceinline__ uint32_t up16_tap (uint32_t p, uint32_t q, uint32_t F, uint32_t G)
{
    if (DOT)
    {
        const uint32_t FG = F | (G << 16);
        const uint32_t lo = pair_dot (q, p, PAIR_LO, FG) >> 8;
        const uint32_t hi = pair_dot (q, p, PAIR_HI, FG) >> 8;
        return lo | (hi << 16);
    }
    const uint32_t lo = (__umul24 (p & 0xffffu, F) + __umul24 (q & 0xffffu, G)) >> 8;
    const uint32_t hi = (__umul24 (p >> 16, F) + __umul24 (q >> 16, G)) >> 8;
    return lo | (hi << 16);
}

template <int INK>
__device__ __forceinline__ void up16_body (const JobDev &J, uint8_t *lds)
{
    constexpr uint32_t TW = TILE_COLS * UP_CPL;

    const uint32_t lane = threadIdx.x;
    const uint32_t tiles_x = (J.w_out + TW - 1) / TW;
    const uint32_t bid = J.tile_band ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;     /* see up_body */
    const uint32_t tx = bid % tiles_x;
    const uint32_t ty = bid / tiles_x;
    const uint32_t yrel0 = ty * J.tile_rows;

    if (yrel0 >= J.n_rows)
        return;

    const uint32_t nrows = (yrel0 + J.tile_rows < J.n_rows) ? J.tile_rows : J.n_rows - yrel0;
    const uint32_t x0 = tx * TW;
    const uint32_t xn = (x0 + TW < J.w_out) ? x0 + TW : J.w_out;
    const uint32_t xl = x0 + lane * UP_CPL;

    uint32_t tap_ofs [UP_CPL], tap_f [UP_CPL];
#pragma unroll
    for (int c = 0; c < UP_CPL; c++)
    {
        const uint32_t xc = (xl + c < xn) ? xl + c : xn - 1;
        bilin_tap (J.frac0_x + (uint64_t) xc * J.step_x, J.w_in, tap_ofs [c], tap_f [c]);
    }
    const uint32_t sx0 = smol_readlane (tap_ofs [0], 0);

    uint32_t my_oy, my_f;
    {
        const uint32_t r = (lane < nrows) ? lane : nrows - 1;
        bilin_tap (J.frac0_y + (uint64_t) (J.first_row + yrel0 + r) * J.step_y, J.h_in, my_oy, my_f);
    }
    const uint32_t oy_first = smol_readlane (my_oy, 0);
    const uint32_t oy_last = smol_readlane (my_oy, nrows - 1);
    const uint32_t sched = (my_oy - oy_first) | (my_f << 16);
    uint32_t n_src = oy_last - oy_first + 2;
    if (n_src > (uint32_t) UP_NSRC_W)
        n_src = UP_NSRC_W;                              /* cannot happen: up_setup sizes tile_rows */

    /* ---- stage: every source pixel of the tile once, as two words of 16-bit channels
     * (R * a | G * a << 16,  B * a | ((a << 8) | 0x80) << 16), 8 bytes per pixel; the
     * inverse table follows the pixels in LDS ------------------------------------- */
    const uint32_t span = J.span_px;
    uint32_t *inv_lut = reinterpret_cast<uint32_t *> (lds + (size_t) UP_NSRC_W * span * 8);
    {
        const gbyte *srow0 = gptr (J.src) + (size_t) oy_first * J.src_stride;
        const uint32_t total = n_src * span;
        const uint32_t sr = 8u * J.pos_in [0], sg = 8u * J.pos_in [1], sb = 8u * J.pos_in [2], sa = 8u * J.pos_in [3];

        for (uint32_t i0 = 0; i0 < total; i0 += 64u * UP_STAGE_UNROLL)
        {
            uint32_t w [UP_STAGE_UNROLL];
#pragma unroll
            for (int u = 0; u < UP_STAGE_UNROLL; u++)
            {
                const uint32_t i = i0 + (uint32_t) u * 64u + lane;
                const uint32_t srow = (uint32_t) (((uint64_t) i * J.span_chunks_inv) >> 32);   /* i / span */
                const uint32_t k = i - srow * span;
                uint32_t x = sx0 + k;
                if (x > J.w_in - 1)
                    x = J.w_in - 1;
                w [u] = 0;
                if (i < total)
                    w [u] = up_load_px<INK> (J, srow0 + (size_t) srow * J.src_stride, x);
            }
#pragma unroll
            for (int u = 0; u < UP_STAGE_UNROLL; u++)
            {
                const uint32_t i = i0 + (uint32_t) u * 64u + lane;
                const uint32_t a = (w [u] >> sa) & 0xffu;
                const uint32_t rg = __umul24 ((w [u] >> sr) & 0xffu, a) | (__umul24 ((w [u] >> sg) & 0xffu, a) << 16);
                const uint32_t ba = __umul24 ((w [u] >> sb) & 0xffu, a) | (((a << 8) | 0x80u) << 16);
                const u32x2a8 pair = { rg, ba };
                if (i < total)
                    *reinterpret_cast<u32x2a8 *> (lds + (size_t) i * 8) = pair;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            inv_lut [u * 64 + lane] = smol_inv_p16_lut [u * 64 + lane];
    }
    smol_wave_sync ();

    /* ---- H ---------------------------------------------------------------------------- */
    uint32_t h_rg [UP_NSRC_W][UP_CPL], h_ba [UP_NSRC_W][UP_CPL];
#pragma unroll
    for (int s = 0; s < UP_NSRC_W; s++)
    {
#pragma unroll
        for (int c = 0; c < UP_CPL; c++)
            h_rg [s][c] = h_ba [s][c] = 0;
        if ((uint32_t) s < n_src)                       /* wave-uniform */
        {
#pragma unroll
            for (int c = 0; c < UP_CPL; c++)
            {
                const uint32_t k = (uint32_t) s * span + (tap_ofs [c] - sx0);
                const u32x4a8 pq = *reinterpret_cast<const u32x4a8 *> (lds + (size_t) k * 8);
                const uint32_t F = tap_f [c], G = 256u - F;
                h_rg [s][c] = up16_tap<true> (pq.x, pq.z, F, G);
                h_ba [s][c] = up16_tap<true> (pq.y, pq.w, F, G);
            }
        }
    }

    /* ---- V + un-premultiply + store -------------------------------------------------------- */
    const bool fast = (xn == x0 + TW) && J.out_vec16_ok;
    /* the destination word from the registers the un-premultiply leaves its byte in (byte 2 of each product) and the
     * alpha byte: two v_perm + one OR (WidePack<WP_P16>'s assembly, smol_kernels.hip.h) */
    const uint32_t sel01 = wide_sel (wide_sel (0x0c0c0c0cu, J.pos_out [0], 4u + 2u), J.pos_out [1], 2u);
    const uint32_t sel2a = wide_sel (wide_sel (0x0c0c0c0cu, J.pos_out [2], 4u + 2u), J.pos_out [3] & 3u, 0u);
    gbyte *out_row = gptr (J.dst) + (size_t) yrel0 * J.dst_stride;
    uint32_t r = 0;
    uint32_t e = smol_readlane (sched, 0);

#pragma unroll
    for (int s = 0; s + 1 < UP_NSRC_W; s++)
    {
        /* rows s and s + 1 of every column side by side, channel by channel (top | bottom << 16): four v_perm per column
         * and row pair, shared by the output rows between them, after which an output pixel's vertical tap is ONE
         * v_dot2_u32_u16 per channel against (F, 256 - F) and comes out unpacked, as the un-premultiply wants it --
         * 17 instructions per pixel where two packed taps, their re-packing, the extractions and the shifts of the
         * assembly took 31 (premul16 x4: 62 -> see profiles/r04_up16_ab.txt) */
        uint32_t pr [UP_CPL][4];
        if (r < nrows && (e & 0xffffu) == (uint32_t) s)
        {
#pragma unroll
            for (int c = 0; c < UP_CPL; c++)
            {
                pr [c][0] = __builtin_amdgcn_perm (h_rg [s + 1][c], h_rg [s][c], PAIR_LO);
                pr [c][1] = __builtin_amdgcn_perm (h_rg [s + 1][c], h_rg [s][c], PAIR_HI);
                pr [c][2] = __builtin_amdgcn_perm (h_ba [s + 1][c], h_ba [s][c], PAIR_LO);
                pr [c][3] = __builtin_amdgcn_perm (h_ba [s + 1][c], h_ba [s][c], PAIR_HI);
            }
        }
        while (r < nrows && (e & 0xffffu) == (uint32_t) s)
        {
            const uint32_t F = e >> 16, G = 256u - F;
            const uint32_t FG = F | (G << 16);
            uint32_t o [UP_CPL];
#pragma unroll
            for (int c = 0; c < UP_CPL; c++)
            {
                const uint32_t vr = pair_word_dot (pr [c][0], FG) >> 8;
                const uint32_t vg = pair_word_dot (pr [c][1], FG) >> 8;
                const uint32_t vb = pair_word_dot (pr [c][2], FG) >> 8;
                const uint32_t a = pair_word_dot (pr [c][3], FG) >> 16;       /* (alpha lane >> 8) & 0xff */
                const uint32_t inv = inv_lut [a];
                /* (s * inv) >> 16 & 0xff (:1036-1100): s < 2^16, inv <= 2^16, the product fits 32 bits; byte 2 of it */
                o [c] = __builtin_amdgcn_perm (__umul24 (vr, inv), __umul24 (vg, inv), sel01)
                      | __builtin_amdgcn_perm (__umul24 (vb, inv), a, sel2a);
            }

            if (fast)
            {
                const u32x4 v = { o [0], o [1], o [2], o [3] };
                /* (round 5: write-back at the run's ends on rows that start anywhere -- premul16 x3 on an odd width 86 -> 75 us,
                 * x4 76 -> 62, aligned rows and x2.6 unchanged: profiles/r05_up_rgb8.txt) */
                up_store16 (J, out_row, xl, lane, v);
            }
            else
            {
#pragma unroll
                for (int c = 0; c < UP_CPL; c++)
                {
                    if (xl + c < xn)
                    {
                        gbyte *p = out_row + (size_t) (xl + c) * 4;
                        if (J.out_dword_ok)
                        {
                            *reinterpret_cast<SMOL_AS_GLOBAL uint32_t *> (p) = o [c];
                        }
                        else
                        {
                            p [0] = (uint8_t) o [c];
                            p [1] = (uint8_t) (o [c] >> 8);
                            p [2] = (uint8_t) (o [c] >> 16);
                            p [3] = (uint8_t) (o [c] >> 24);
                        }
                    }
                }
            }
            out_row += J.dst_stride;
            r++;
            if (r < nrows)
                e = smol_readlane (sched, r);
        }
    }
}

/* ------------------------------------------------------------------------- *
 * k_up, table-driven flavours: every other 128bpp-class magnification --      *
 * sRGB-linear (with_srgb = 1, gamma-correct scaling; 11-bit channels, W32 = false)  *
 * and premul16 + sRGB (19-bit channels, W32 = true) -- any pixel types, any   *
 * alignment. Pixels go through WideUnpack / WidePack (smol_kernels.hip.h),    *
 * flavour chosen per batch (smolscale-generic.c:470-752, 912-1164); 4 waves per workgroup share one    *
 * LDS copy of the six tables, each wave works on its own tile.                *
 * ------------------------------------------------------------------------- */
constexpr int UPG_NSRC = SMOL_UPG_NSRC;     /* source rows per tile of the table-driven flavours (smol_jobdev.h) */

template <int I, int N, typename F>
__device__ __forceinline__ void up_static_for (F &&f)
{
    if constexpr (I < N)
    {
        f (std::integral_constant<int, I> {});
        up_static_for<I + 1, N> (f);
    }
}

template <bool W32>
__device__ __forceinline__ void upg_body (const JobDev &J, uint8_t *lds, const uint8_t *lut_lds,
                                          const uint32_t lane, const uint32_t tile)
{
    constexpr int CPL = W32 ? 2 : 4;
    constexpr uint32_t TW = TILE_COLS * CPL;
    constexpr int NCH = W32 ? 4 : 2;                    /* registers per (source row, column) */

    const uint32_t tiles_x = (J.w_out + TW - 1) / TW;
    const uint32_t tx = tile % tiles_x;
    const uint32_t ty = tile / tiles_x;
    const uint32_t yrel0 = ty * J.tile_rows;

    if (yrel0 >= J.n_rows)
        return;

    const uint32_t nrows = (yrel0 + J.tile_rows < J.n_rows) ? J.tile_rows : J.n_rows - yrel0;
    const uint32_t x0 = tx * TW;
    const uint32_t xn = (x0 + TW < J.w_out) ? x0 + TW : J.w_out;
    const uint32_t xl = x0 + lane * CPL;
    const LutLds lut (lut_lds, LUT_AT_LDS_0);       /* (upg_entry: lut_expect_lds_0) */
    const int uk = wide_unpack_kind (J), pk = wide_pack_kind (J);

    uint32_t tap_ofs [CPL], tap_f [CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++)
    {
        const uint32_t xc = (xl + c < xn) ? xl + c : xn - 1;
        bilin_tap (J.frac0_x + (uint64_t) xc * J.step_x, J.w_in, tap_ofs [c], tap_f [c]);
    }
    const uint32_t sx0 = smol_readlane (tap_ofs [0], 0);
    Lerp11W tap_w [CPL];                                /* W32 = false: 11-bit channels, packed taps (lerp11_pk) */
#pragma unroll
    for (int c = 0; c < CPL; c++)
        tap_w [c] = lerp11_weights (tap_f [c]);

    uint32_t my_oy, my_f;
    {
        const uint32_t r = (lane < nrows) ? lane : nrows - 1;
        bilin_tap (J.frac0_y + (uint64_t) (J.first_row + yrel0 + r) * J.step_y, J.h_in, my_oy, my_f);
    }
    const uint32_t oy_first = smol_readlane (my_oy, 0);
    const uint32_t oy_last = smol_readlane (my_oy, nrows - 1);
    const uint32_t sched = (my_oy - oy_first) | (my_f << 16);
    uint32_t n_src = oy_last - oy_first + 2;
    if (n_src > (uint32_t) UPG_NSRC)
        n_src = UPG_NSRC;                              /* cannot happen: up_setup sizes tile_rows */

    /* ---- stage: every source pixel of the tile once, as internal channels ------------------ */
    const uint32_t span = J.span_px;
    {
        const gbyte *srow0 = gptr (J.src) + (size_t) oy_first * J.src_stride;
        const uint32_t total = n_src * span;

        for (uint32_t i0 = 0; i0 < total; i0 += 64u * 4u)
        {
            uint32_t w [4];
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                const uint32_t i = i0 + (uint32_t) u * 64u + lane;
                const uint32_t srow = (uint32_t) (((uint64_t) i * J.span_chunks_inv) >> 32);   /* i / span */
                const uint32_t k = i - srow * span;
                uint32_t x = sx0 + k;
                if (x > J.w_in - 1)
                    x = J.w_in - 1;
                w [u] = 0;
                if (i < total)
                {
                    if (J.in_dword_ok)
                        w [u] = up_load_px<UP_IN_A4> (J, srow0 + (size_t) srow * J.src_stride, x);
                    else
                        w [u] = up_load_px<UP_IN_ANY> (J, srow0 + (size_t) srow * J.src_stride, x);
                }
            }
            /* conversion flavour chosen once per batch of four pixels per lane (WideUnpack: branch-free inside) */
            auto conv = [&] (const auto &un)
            {
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const uint32_t i = i0 + (uint32_t) u * 64u + lane;
                    uint32_t ch [4];
                    un (w [u], ch);
                    if (i < total)
                    {
                        if (W32)
                            *reinterpret_cast<u32x4 *> (lds + (size_t) i * 16) = u32x4 { ch [0], ch [1], ch [2], ch [3] };
                        else
                            *reinterpret_cast<u32x2a8 *> (lds + (size_t) i * 8) = u32x2a8 { ch [0] | (ch [1] << 16), ch [2] | (ch [3] << 16) };
                    }
                }
            };
            if (W32 && uk == WU_P16L)
                conv (WideUnpack<WU_P16L> (J, lut));
            else if (!W32 && (uk == WU_LIN_P || uk == WU_LIN_U))
            {
                uint32_t all = 0xffffffffu;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    all &= (i0 + (uint32_t) u * 64u + lane < total) ? w [u] : 0xffffffffu;
                if (wide_batch_opaque (J, all))
                    conv (WideUnpack<WU_LIN_O> (J, lut));
                else if (uk == WU_LIN_P)
                    conv (WideUnpack<WU_LIN_P> (J, lut));
                else
                    conv (WideUnpack<WU_LIN_U> (J, lut));
            }
            else if (!W32 && uk == WU_LIN_O)
                conv (WideUnpack<WU_LIN_O> (J, lut));
            else
                conv (WideUnpack<WU_ANY> (J, lut));
        }
    }
    smol_wave_sync ();

    /* ---- H ---------------------------------------------------------------------------- */
    uint32_t h [UPG_NSRC][CPL][NCH];
#pragma unroll
    for (int s = 0; s < UPG_NSRC; s++)
    {
#pragma unroll
        for (int c = 0; c < CPL; c++)
#pragma unroll
            for (int i = 0; i < NCH; i++)
                h [s][c][i] = 0;
        if ((uint32_t) s < n_src)                       /* wave-uniform */
        {
#pragma unroll
            for (int c = 0; c < CPL; c++)
            {
                const uint32_t k = (uint32_t) s * span + (tap_ofs [c] - sx0);
                const uint32_t F = tap_f [c], G = 256u - F;
                if (W32)
                {
                    const u32x4 p = *reinterpret_cast<const u32x4 *> (lds + (size_t) k * 16);
                    const u32x4 q = *reinterpret_cast<const u32x4 *> (lds + (size_t) k * 16 + 16);
                    h [s][c][0] = (__umul24 (p.x, F) + __umul24 (q.x, G)) >> 8;
                    h [s][c][1] = (__umul24 (p.y, F) + __umul24 (q.y, G)) >> 8;
                    h [s][c][2 % NCH] = (__umul24 (p.z, F) + __umul24 (q.z, G)) >> 8;
                    h [s][c][3 % NCH] = (__umul24 (p.w, F) + __umul24 (q.w, G)) >> 8;
                }
                else
                {
                    const u32x4a8 pq = *reinterpret_cast<const u32x4a8 *> (lds + (size_t) k * 8);
                    h [s][c][0] = lerp11_pk (pq.x, pq.z, tap_w [c]);
                    h [s][c][1] = lerp11_pk (pq.y, pq.w, tap_w [c]);
                }
            }
        }
    }

    /* ---- V + pack + store -------------------------------------------------------------------- */
    uint8_t *out_row = J.dst + (size_t) yrel0 * J.dst_stride;
    uint32_t r = 0;
    uint32_t e = smol_readlane (sched, 0);
    /* full-width tile of 32-bit pixels on a 16-byte aligned destination (wave-uniform) */
    const bool vec_ok = (J.bpp_out == 4) && (xn == x0 + TW) && J.out_vec16_ok;

    /* compile-time loop over the source-row pairs: h [] must be indexed by constants (a loop the
     * optimiser declines to unroll would put it in scratch memory) */
    up_static_for<0, UPG_NSRC - 1> ([&] (auto s_const)
    {
        constexpr int s = decltype (s_const)::value;
        /* 11-bit channels: the rows s and s + 1 of every column side by side, channel by channel -- (top | bottom << 16),
         * four v_perm per column and row pair -- so that an output pixel's vertical tap is ONE v_dot2_u32_u16 per channel
         * against the packed weights (F, 256 - F), (top * F + bottom * (256 - F)) >> 8, and comes out unpacked, as the
         * packer wants it: 4 dot products + 4 shifts per pixel where two packed taps and their unpacking took 16
         * instructions (gamma-correct x4: 102 -> see profiles/r04_upg_dot2_ab.txt) */
        uint32_t pr [CPL][4];
        if (!W32 && r < nrows && (e & 0xffffu) == (uint32_t) s)
        {
#pragma unroll
            for (int c = 0; c < CPL; c++)
            {
                pr [c][0] = __builtin_amdgcn_perm (h [s + 1][c][0], h [s][c][0], PAIR_LO);
                pr [c][1] = __builtin_amdgcn_perm (h [s + 1][c][0], h [s][c][0], PAIR_HI);
                pr [c][2] = __builtin_amdgcn_perm (h [s + 1][c][1 % NCH], h [s][c][1 % NCH], PAIR_LO);
                pr [c][3] = __builtin_amdgcn_perm (h [s + 1][c][1 % NCH], h [s][c][1 % NCH], PAIR_HI);
            }
        }
        while (r < nrows && (e & 0xffffu) == (uint32_t) s)
        {
            const uint32_t F = e >> 16, G = 256u - F;
            const uint32_t FG = F | (G << 16);
            uint32_t wv [CPL], vv [CPL][4];
#pragma unroll
            for (int c = 0; c < CPL; c++)
            {
                if (W32)
                {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        vv [c][i] = (__umul24 (h [s][c][i % NCH], F) + __umul24 (h [s + 1][c][i % NCH], G)) >> 8;
                }
                else
                {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        vv [c][i] = pair_word_dot (pr [c][i], FG) >> 8;
                }
            }
            /* pack: flavour chosen once per output row and wave (WidePack: the row's 2-4 pixels per lane convert in one
             * branch-free block); all-opaque rows skip the inverse-alpha look-up */
            auto put = [&] (const auto &pack)
            {
#pragma unroll
                for (int c = 0; c < CPL; c++)
                {
                    wv [c] = pack (vv [c]);
                    if (!vec_ok && xl + c < xn)
                        wide_store_word (J, gptr (out_row), xl + c, wv [c]);
                }
            };
            if (W32 && pk == WP_P16L)
                put (WidePack<WP_P16L> (J, lut));
            else if (!W32 && (pk == WP_LIN_P || pk == WP_LIN_U))
            {
                uint32_t all = 0xffu;
#pragma unroll
                for (int c = 0; c < CPL; c++)
                    all &= (xl + c < xn) ? vv [c][3] : 0xffu;
                if (smol_wave_all_opaque (all & 0xffu))
                    put (WidePack<WP_LIN_O> (J, lut));
                else if (pk == WP_LIN_P)
                    put (WidePack<WP_LIN_P> (J, lut));
                else
                    put (WidePack<WP_LIN_U> (J, lut));
            }
            else if (!W32 && pk == WP_LIN_Q)
                put (WidePack<WP_LIN_Q> (J, lut));
            else
                put (WidePack<WP_ANY> (J, lut));
            if (vec_ok)
            {
                /* the lane's adjacent pixels leave as ONE non-temporal store (16 or 8 bytes) */
                if (CPL == 4)
                {
                    const u32x4 v4 = { wv [0], wv [1 % CPL], wv [2 % CPL], wv [3 % CPL] };
                    /* (up_store16's write-back at the run's ends costs this flavour 6 % on aligned rows -- it lives at 127
                     * registers --: 95 -> 101 us, profiles/r05_up_rgb8.txt) */
                    store_nt128 (gptr (out_row) + (size_t) xl * 4, v4);
                }
                else
                {
                    const u32x2 v2 = { wv [0], wv [1 % CPL] };
#ifdef SMOL_KERNEL_EMU
                    __builtin_memcpy (gptr (out_row) + (size_t) xl * 4, &v2, 8);        /* dword-aligned only */
#else
                    typedef u32x2 u32x2a4 __attribute__ ((aligned (4)));
                    __builtin_nontemporal_store (v2, reinterpret_cast<SMOL_AS_GLOBAL u32x2a4 *> (gptr (out_row) + (size_t) xl * 4));
#endif
                }
            }
            out_row += J.dst_stride;
            r++;
            if (r < nrows)
                e = smol_readlane (sched, r);
        }
    });
}

template <bool W32>
__device__ __forceinline__ void upg_entry (const JobDev &J, uint8_t *lds)
{
#ifdef SMOL_KERNEL_EMU
    static uint8_t lut_lds [LUT_LDS_BYTES];
    lut_fill_lds (lut_lds, 0, 1);                       /* emulated: one wave per block */
    upg_body<W32> (J, lds, lut_lds, threadIdx.x, blockIdx.x);
#else
    const uint32_t wave = smol_wave_index ();       /* a SCALAR: everything derived from the tile index stays wave-uniform */
    lut_expect_lds_0 (lds);
    lut_fill_lds (lds, threadIdx.x, 256);
    __syncthreads ();
    const uint32_t region = (UPG_NSRC * J.span_px * (W32 ? 16u : 8u) + 15u) & ~15u;
    upg_body<W32> (J, lds + ((LUT_LDS_BYTES + 15u) & ~15u) + wave * region, lds,
                   threadIdx.x & 63u, blockIdx.x * 4u + wave);
#endif
}

#ifdef SMOL_KERNEL_EMU
/* host emulation: one entry that switches at run time */
__global__ void k_up_one (const JobDev J)
{
    const bool a4 = J.tile_aligned4;
    if (J.up_wide >= 2)
    {
        if (J.up_wide == 3) upg_entry<true> (J, smol_dyn_lds); else upg_entry<false> (J, smol_dyn_lds);
        return;
    }
    if (J.up_wide)
    {
        if (a4) up16_body<UP_IN_A4> (J, smol_dyn_lds); else up16_body<UP_IN_ANY> (J, smol_dyn_lds);
        return;
    }
    switch (J.tile_variant & 3u)
    {
        case OUT_P4: if (a4) up_body<UP_IN_A4, OUT_P4> (J, smol_dyn_lds); else up_body<UP_IN_ANY, OUT_P4> (J, smol_dyn_lds); break;
        case OUT_U4: if (a4) up_body<UP_IN_A4, OUT_U4> (J, smol_dyn_lds); else up_body<UP_IN_ANY, OUT_U4> (J, smol_dyn_lds); break;
        default:     if (a4) up_body<UP_IN_A4, OUT_ANY> (J, smol_dyn_lds); else up_body<UP_IN_ANY, OUT_ANY> (J, smol_dyn_lds); break;
    }
}
#else
template <int INK, int OUT>
__global__ void __launch_bounds__ (64)
k_up_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.y);
    if (OUT == OUT_W16)
        up16_body<INK> (J, smol_dyn_lds);
    else
        up_body<INK, OUT == OUT_W16 ? OUT_P4 : OUT> (J, smol_dyn_lds);
}

/* registers against resident waves (profiles/r03i_upg_occ_ab.txt): the 19-bit flavour (99 registers as compiled) gains
 * 12 % when held to 6 waves per SIMD (premul16 + gamma-correct x4: 100 -> 88 us); the 11-bit flavour, four columns per
 * lane and 127 registers, loses when squeezed (5 waves 112 -> 116 us, 6 waves 167: spills) */
#define SMOL_UPG_OCC(W32) __attribute__ ((amdgpu_waves_per_eu ((W32) ? 6 : 4)))

template <bool W32>
__global__ void __launch_bounds__ (256) SMOL_UPG_OCC (W32)
k_upg_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.y);
    upg_entry<W32> (J, smol_dyn_lds);
}

template <bool W32>
__global__ void __launch_bounds__ (256) SMOL_UPG_OCC (W32)
k_upg_one (const JobDev J)
{
    upg_entry<W32> (J, smol_dyn_lds);
}

template <int INK, int OUT>
__global__ void __launch_bounds__ (64)
k_up_one (const JobDev J)
{
    if (OUT == OUT_W16)
        up16_body<INK> (J, smol_dyn_lds);
    else
        up_body<INK, OUT == OUT_W16 ? OUT_P4 : OUT> (J, smol_dyn_lds);
}
#endif

} /* namespace smolhip */
