/* smol_kernel_down.hip.h -- k_down: bilinear MINIFICATION on both axes (0-2
 * halvings each) for the premul16 class without sRGB -- unassociated in AND out,
 * the 128bpp-class default for RGBA8 data (smolscale-generic.c:522-560 unpack,
 * :1036-1100 pack) -- on 4-byte aligned rows. Included at the end of
 * smol_kernels.hip.h. (The same structure was tried for the 64bpp class and lost
 * to k_tile by 10-40 %: DESIGN.md section 4.)
 *
 * One wave owns 64 (or, without horizontal halvings, 256) output columns x tile_rows
 * output rows; a lane owns one column (or 4 adjacent ones: 16-byte stores).
 *
 *   stage  source rows stream through in order. Row r+1 is loaded while row r is
 *          filtered; on arrival each pixel is converted ONCE to its internal
 *          channels and written to one of two LDS row buffers (8 or 16 bytes per
 *          pixel).
 *   H      a lane reads the pixel pair of each of its 2^HX samples with one 16-byte
 *          LDS read and blends it (:1290-1425);
 *   V      the vertical filter runs in registers per column against the previous
 *          row's result (:1648-2007), following a schedule the lanes computed once
 *          (read back with v_readlane); an output pixel is un-premultiplied through
 *          the inverse table (LDS copy) and leaves as soon as its last row has passed.
 *
 * Values are 16 bit, two per register, taps are 24-bit multiplies:
 *      q + (((p - q) * F) >> 8)  ==  (p * F + q * (256 - F)) >> 8
 * (an integer identity, both sides below 2^24). */

#pragma once

#include "smol_kernel_lind.hip.h"

namespace smolhip {

constexpr int DOWN_NL = 3;          /* 16-byte loads per lane per source row, at most (span <= 768 pixels) */

typedef uint32_t u32x4a4 __attribute__ ((ext_vector_type (4), aligned (4)));

/* MODE: how a pixel is converted and kept
 *   DOWN_P16   premul16 without sRGB on aligned 32-bit rows: table-free unpack, 16-bit channel pairs
 *   DOWN_GEN16 any other 128bpp-class job whose channels fit 16 bits (sRGB-linear with 8-bit alpha:
 *              11-bit values), any input type / alignment: unpack_word<true> / store_px<true>
 *   DOWN_GEN32 premul16 + sRGB-linear (19-bit values): four 32-bit channels per pixel */
enum : int { DOWN_P16 = 0, DOWN_GEN16 = 1, DOWN_GEN32 = 2 };

template <int HX, int CPL, int MODE>
__device__ __forceinline__ void down_body (const JobDev &J, uint8_t *lds, const uint8_t *lut_lds,
                                           const uint32_t lane, const uint32_t tile)
{
    constexpr uint32_t NSX = 1u << HX;
    constexpr uint32_t TW = TILE_COLS * CPL;            /* output columns per tile */
    constexpr uint32_t PXB = (MODE == DOWN_GEN32) ? 16u : 8u;      /* LDS bytes per staged pixel */
    constexpr bool GEN = (MODE != DOWN_P16);
    static_assert (!GEN || CPL == 1 || (MODE == DOWN_GEN16 && HX == 0), "table-driven flavours: one column per lane, or four without horizontal halvings (11-bit channels)");

    const uint32_t tiles_x = (J.w_out + TW - 1) / TW;
    const uint32_t tx = tile % tiles_x;
    const uint32_t ty = tile / tiles_x;
    const uint32_t yrel0 = ty * J.tile_rows;

    if (yrel0 >= J.n_rows)
        return;

    const uint32_t yrel1 = (yrel0 + J.tile_rows < J.n_rows) ? yrel0 + J.tile_rows : J.n_rows;
    const uint32_t x0 = tx * TW;
    const uint32_t xn = (x0 + TW < J.w_out) ? x0 + TW : J.w_out;
    const uint32_t xl = x0 + lane * CPL;                            /* this lane's first column */

    /* ---- horizontal samples of this lane's columns ------------------------------ */
    uint32_t tap_ofs [CPL][NSX], tap_f [CPL][NSX];
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
    const uint32_t sx0 = smol_readlane (tap_ofs [0][0], 0);         /* first source pixel of the tile */
    uint32_t sx1, f_unused;
    bilin_tap (J.frac0_x + (uint64_t) ((xn << HX) - 1) * J.step_x, J.w_in, sx1, f_unused);
    sx1 += 1;                                                       /* last source pixel of the tile  */
    const uint32_t nch = (sx1 - sx0 + 4) >> 2;                      /* 4-pixel chunks per staged row  */
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (uint32_t s = 0; s < NSX; s++)
            tap_ofs [c][s] = (tap_ofs [c][s] - sx0) * PXB;          /* byte offset in the LDS row     */
    Lerp11W tap_w [CPL][NSX];                                       /* DOWN_GEN16: 11-bit channels, packed taps (lerp11_pk) */
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (uint32_t s = 0; s < NSX; s++)
            tap_w [c][s] = lerp11_weights (tap_f [c][s]);
    if (MODE == DOWN_P16)
    {
#pragma unroll
        for (int c = 0; c < CPL; c++)
#pragma unroll
            for (uint32_t s = 0; s < NSX; s++)
                tap_f [c][s] |= (256u - tap_f [c][s]) << 16;        /* (F, 256 - F) as one packed operand */
    }

    /* ---- vertical schedule (as in k_tile): sample k of the tile in lane k & 63 of sched [k >> 6] */
    const uint32_t hv = J.halv_v;
    const uint32_t nsy = 1u << hv;
    const uint32_t jy0 = (J.first_row + yrel0) << hv;
    const uint32_t n_samples = (yrel1 - yrel0) << hv;               /* <= 128 (down_setup) */
    uint32_t sched [2];
#pragma unroll
    for (uint32_t q = 0; q < 2; q++)
    {
        uint32_t k = q * 64u + lane;
        if (k >= n_samples)
            k = n_samples - 1;
        uint32_t o, f;
        bilin_tap (J.frac0_y + (uint64_t) (jy0 + k) * J.step_y, J.h_in, o, f);
        sched [q] = o | (f << 16);
    }
    uint32_t e = smol_readlane (sched [0], 0);
    uint32_t oy = e & 0xffffu, Fy = e >> 16;
    const uint32_t row_first = oy;
    const uint32_t e_last = smol_readlane (sched [(n_samples - 1) >> 6], (n_samples - 1) & 63u);
    const uint32_t row_last = (e_last & 0xffffu) + 1;

    /* ---- staging ---------------------------------------------------------------------------- */
    const uint32_t buf_pitch = (J.span_px * PXB + 15u) & ~15u;      /* one LDS row buffer (host bound on the span) */
    uint32_t *inv_lut = reinterpret_cast<uint32_t *> (lds + 2 * (size_t) buf_pitch);       /* DOWN_P16 only */
    /* table-driven flavours only; DOWN_P16 has no table copy and passes nullptr (never dereferenced -- the host build, which
     * UBSan checks, does not offset it either) */
#ifdef SMOL_KERNEL_EMU
    const LutLds lut (GEN ? lut_lds : lds);
#else
    const LutLds lut (lut_lds);
#endif
    const uint32_t sr = 8u * J.pos_in [0], sg = 8u * J.pos_in [1], sb = 8u * J.pos_in [2], sa = 8u * J.pos_in [3];
    const int uk = GEN ? wide_unpack_kind (J) : WU_P16, pk = GEN ? wide_pack_kind (J) : WP_P16;
    const gbyte *srow = gptr (J.src) + (size_t) row_first * J.src_stride;   /* row being loaded */

    /* DOWN_P16 moves 4-pixel chunks (16-byte loads, cheap conversion); the table-driven
     * flavours spread single pixels over all 64 lanes, because their conversion is the
     * expensive part and a short span would otherwise keep most lanes idle */
    constexpr int NRAW = GEN ? (2 << HX) * CPL + 1 : DOWN_NL;    /* a tile of 64 * CPL columns spans < 64 * CPL * 2^(HX+1) + 3 pixels */
    const uint32_t npx = sx1 - sx0 + 1;                             /* pixels per staged row */
    u32x4 raw [GEN ? 1 : DOWN_NL];
    uint32_t rawpx [GEN ? (2 << HX) * CPL + 1 : 1];

    auto load_row = [&] ()
    {
        if (GEN)
        {
#pragma unroll
            for (int l = 0; l < NRAW; l++)
            {
                const uint32_t i = (uint32_t) l * 64u + lane;
                rawpx [l] = 0;
                if ((uint32_t) l * 64u < npx && i < npx)            /* first test is wave-uniform */
                {
                    if (J.in_dword_ok)
                        rawpx [l] = *reinterpret_cast<const SMOL_AS_GLOBAL uint32_t *> (srow + (size_t) (sx0 + i) * 4);
                    else
                        rawpx [l] = up_load_px<UP_IN_ANY> (J, srow, sx0 + i);
                }
            }
            srow += J.src_stride;
            return;
        }
#pragma unroll
        for (int l = 0; l < DOWN_NL; l++)
        {
            const uint32_t c = (uint32_t) l * 64u + lane;
            raw [l] = u32x4 { 0, 0, 0, 0 };
            if ((uint32_t) l * 64u < nch && c < nch)                /* first test is wave-uniform */
            {
                const uint32_t k = sx0 + 4u * c;
                if (k + 3u <= J.w_in - 1u)
                {
                    raw [l] = *reinterpret_cast<const SMOL_AS_GLOBAL u32x4a4 *> (srow + (size_t) k * 4);
                }
                else
                {
                    /* the row's last pixels: never read past the end of the row */
                    const SMOL_AS_GLOBAL uint32_t *px = reinterpret_cast<const SMOL_AS_GLOBAL uint32_t *> (srow);
                    const uint32_t last = J.w_in - 1u;
                    raw [l].x = px [k < last ? k : last];
                    raw [l].y = px [k + 1u < last ? k + 1u : last];
                    raw [l].z = px [k + 2u < last ? k + 2u : last];
                    raw [l].w = px [last];
                }
            }
        }
        srow += J.src_stride;
    };

    auto convert_row = [&] (uint8_t *buf)
    {
        if (GEN)
        {
            /* the conversion's flavour is chosen once per row and wave (WideUnpack: branch-free inside, the look-ups
             * of a lane's pixels in flight together); a row whose pixels are all opaque takes the fused table */
            auto conv = [&] (const auto &un)
            {
#pragma unroll
                for (int l = 0; l < NRAW; l++)
                {
                    const uint32_t i = (uint32_t) l * 64u + lane;
                    if ((uint32_t) l * 64u < npx && i < npx)
                    {
                        uint32_t ch [4];
                        un (rawpx [l], ch);
                        if (MODE == DOWN_GEN32)
                            *reinterpret_cast<u32x4 *> (buf + (size_t) i * 16) = u32x4 { ch [0], ch [1], ch [2], ch [3] };
                        else
                            *reinterpret_cast<u32x2a8 *> (buf + (size_t) i * 8) = u32x2a8 { ch [0] | (ch [1] << 16), ch [2] | (ch [3] << 16) };
                    }
                }
            };
            if (uk == WU_LIN_P || uk == WU_LIN_U)
            {
                uint32_t all = 0xffffffffu;
#pragma unroll
                for (int l = 0; l < NRAW; l++)
                    all &= ((uint32_t) l * 64u + lane < npx) ? rawpx [l] : 0xffffffffu;
                if (wide_batch_opaque (J, all))
                    conv (WideUnpack<WU_LIN_O> (J, lut));
                else if (uk == WU_LIN_P)
                    conv (WideUnpack<WU_LIN_P> (J, lut));
                else
                    conv (WideUnpack<WU_LIN_U> (J, lut));
            }
            else if (uk == WU_LIN_O)
                conv (WideUnpack<WU_LIN_O> (J, lut));
            else if (uk == WU_P16L)
                conv (WideUnpack<WU_P16L> (J, lut));
            else
                conv (WideUnpack<WU_ANY> (J, lut));
            return;
        }
#pragma unroll
        for (int l = 0; l < DOWN_NL; l++)
        {
            const uint32_t c = (uint32_t) l * 64u + lane;
            if ((uint32_t) l * 64u < nch && c < nch)
            {
                const uint32_t w [4] = { raw [l].x, raw [l].y, raw [l].z, raw [l].w };
                uint32_t o [8];
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const uint32_t a = (w [i] >> sa) & 0xffu;
                    o [2 * i] = __umul24 ((w [i] >> sr) & 0xffu, a) | (__umul24 ((w [i] >> sg) & 0xffu, a) << 16);
                    o [2 * i + 1] = __umul24 ((w [i] >> sb) & 0xffu, a) | (((a << 8) | 0x80u) << 16);
                }
                u32x4 *d = reinterpret_cast<u32x4 *> (buf + (size_t) c * 32);
                d [0] = u32x4 { o [0], o [1], o [2], o [3] };
                d [1] = u32x4 { o [4], o [5], o [6], o [7] };
            }
        }
    };

    if (!GEN)
    {
#pragma unroll
        for (int u = 0; u < 4; u++)
            inv_lut [u * 64 + lane] = smol_inv_p16_lut [u * 64 + lane];
    }

    /* ---- the row loop --------------------------------------------------------------------- */
    const uint32_t dr = 8u * J.pos_out [0], dg = 8u * J.pos_out [1], db = 8u * J.pos_out [2], da = 8u * J.pos_out [3];
    const bool interior = (xn == x0 + TW) && J.out_vec16_ok;
    gbyte *out_row = gptr (J.dst) + (size_t) yrel0 * J.dst_stride;
    uint32_t ks = 0, s_y = 0;
    /* previous row's horizontal result (two packed channel pairs; four 32-bit channels
     * in hp32 for DOWN_GEN32) and the vertical sums */
    uint32_t hp0 [CPL], hp1 [CPL], acc [CPL][4], hp32 [4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int c = 0; c < CPL; c++)
    {
        hp0 [c] = hp1 [c] = 0;
        acc [c][0] = acc [c][1] = acc [c][2] = acc [c][3] = 0;
    }

    load_row ();

    for (uint32_t r = row_first; r <= row_last; r++)
    {
        uint8_t *buf = lds + ((r - row_first) & 1u) * buf_pitch;

        convert_row (buf);                              /* waits for row r's loads */
        smol_wave_sync ();
        if (r < row_last)
            load_row ();                                /* row r + 1 flies while row r is filtered */

        /* ---- H ---------------------------------------------------------------------------- */
        uint32_t h0 [CPL], h1 [CPL], h32 [4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int c = 0; c < CPL; c++)
        {
            uint32_t hs [4] = { 0, 0, 0, 0 };
            if (MODE == DOWN_GEN32)
            {
#pragma unroll
                for (uint32_t s = 0; s < NSX; s++)
                {
                    const u32x4 p = *reinterpret_cast<const u32x4 *> (buf + tap_ofs [c][s]);
                    const u32x4 q = *reinterpret_cast<const u32x4 *> (buf + tap_ofs [c][s] + 16);
                    const uint32_t F = tap_f [c][s], G = 256u - F;
                    hs [0] += (__umul24 (p.x, F) + __umul24 (q.x, G)) >> 8;
                    hs [1] += (__umul24 (p.y, F) + __umul24 (q.y, G)) >> 8;
                    hs [2] += (__umul24 (p.z, F) + __umul24 (q.z, G)) >> 8;
                    hs [3] += (__umul24 (p.w, F) + __umul24 (q.w, G)) >> 8;
                }
#pragma unroll
                for (int i = 0; i < 4; i++)
                    h32 [i] = hs [i] >> HX;
                h0 [c] = h1 [c] = 0;
                continue;
            }
#pragma unroll
            for (uint32_t s = 0; s < NSX; s++)
            {
                const u32x4a8 pq = *reinterpret_cast<const u32x4a8 *> (buf + tap_ofs [c][s]);
                if (MODE == DOWN_P16)
                {
                    /* p*F + q*(256-F) per channel as ONE v_dot2_u32_u16 on the (p, q) pair a v_perm puts
                     * side by side, against the packed weights (F, 256 - F). Measured: premul16 ratio 1.5
                     * 61 -> 57.6 us; the table-driven flavours LOSE 2-8 % with it (profiles/r02u_down_dot2.txt) */
                    const uint32_t FG = tap_f [c][s];
                    const uint32_t d0 = pair_dot (pq.z, pq.x, PAIR_LO, FG) >> 8;
                    const uint32_t d1 = pair_dot (pq.z, pq.x, PAIR_HI, FG) >> 8;
                    const uint32_t d2 = pair_dot (pq.w, pq.y, PAIR_LO, FG) >> 8;
                    const uint32_t d3 = pair_dot (pq.w, pq.y, PAIR_HI, FG) >> 8;
                    hs [0] = s ? hs [0] + d0 : d0;
                    hs [1] = s ? hs [1] + d1 : d1;
                    hs [2] = s ? hs [2] + d2 : d2;
                    hs [3] = s ? hs [3] + d3 : d3;
                    continue;
                }
                /* DOWN_GEN16: both channel pairs stay packed; the sums of 2^HX taps stay below 2^13 per lane */
                const uint32_t t01 = lerp11_pk (pq.x, pq.z, tap_w [c][s]);
                const uint32_t t23 = lerp11_pk (pq.y, pq.w, tap_w [c][s]);
                hs [0] = s ? hs [0] + t01 : t01;
                hs [2] = s ? hs [2] + t23 : t23;
            }
            if (MODE == DOWN_GEN16)
            {
                h0 [c] = (hs [0] >> HX) & (0xffffu >> HX) * 0x00010001u;
                h1 [c] = (hs [2] >> HX) & (0xffffu >> HX) * 0x00010001u;
                continue;
            }
            h0 [c] = (hs [0] >> HX) | ((hs [1] >> HX) << 16);
            h1 [c] = (hs [2] >> HX) | ((hs [3] >> HX) << 16);
        }

        /* ---- V: every sample whose bottom row is r ----------------------------------------- */
        while (ks < n_samples && oy + 1u == r)
        {
            const uint32_t G = 256u - Fy;
#pragma unroll
            for (int c = 0; c < CPL; c++)
            {
                if (MODE == DOWN_GEN32)
                {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        const uint32_t t = (__umul24 (hp32 [i], Fy) + __umul24 (h32 [i], G)) >> 8;
                        acc [c][i] = s_y ? acc [c][i] + t : t;
                    }
                    continue;
                }
                uint32_t t0, t1, t2, t3;
                if (MODE == DOWN_P16)
                {
                    const uint32_t FGy = Fy | (G << 16);
                    t0 = pair_dot (h0 [c], hp0 [c], PAIR_LO, FGy) >> 8;
                    t1 = pair_dot (h0 [c], hp0 [c], PAIR_HI, FGy) >> 8;
                    t2 = pair_dot (h1 [c], hp1 [c], PAIR_LO, FGy) >> 8;
                    t3 = pair_dot (h1 [c], hp1 [c], PAIR_HI, FGy) >> 8;
                }
                else
                {
                    /* DOWN_GEN16: packed pairs in acc [c][0] (R, G) and acc [c][2] (B, A) */
                    const Lerp11W wy = lerp11_weights (Fy);
                    t0 = lerp11_pk (hp0 [c], h0 [c], wy);
                    t2 = lerp11_pk (hp1 [c], h1 [c], wy);
                    t1 = t3 = 0;
                }
                acc [c][0] = s_y ? acc [c][0] + t0 : t0;
                acc [c][1] = s_y ? acc [c][1] + t1 : t1;
                acc [c][2] = s_y ? acc [c][2] + t2 : t2;
                acc [c][3] = s_y ? acc [c][3] + t3 : t3;
            }
            s_y++;
            ks++;
            if (s_y == nsy && GEN)
            {
                uint32_t v [CPL][4], all_a = 255u;
#pragma unroll
                for (int c = 0; c < CPL; c++)
                {
                    if (MODE == DOWN_GEN16)
                    {
                        v [c][0] = (acc [c][0] & 0xffffu) >> hv; v [c][1] = acc [c][0] >> (16u + hv);
                        v [c][2] = (acc [c][2] & 0xffffu) >> hv; v [c][3] = acc [c][2] >> (16u + hv);
                    }
                    else
                    {
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            v [c][i] = acc [c][i] >> hv;
                    }
                    all_a &= (xl + c < xn) ? v [c][3] & 0xffu : 255u;
                }
                const bool vec = (CPL == 4) && interior;
                auto put = [&] (const auto &pack)
                {
                    uint32_t w [CPL];
#pragma unroll
                    for (int c = 0; c < CPL; c++)
                    {
                        w [c] = pack (v [c]);
                        if (!vec && xl + c < xn)
                            wide_store_word (J, out_row, xl + c, w [c]);
                    }
                    if (vec)
                        store_row128 (out_row + (size_t) xl * 4, u32x4 { w [0], w [1 % CPL], w [2 % CPL], w [3 % CPL] }, J.out_wt);
                };
                /* once per output row and wave; all-opaque rows skip the inverse-alpha look-up and multiplies */
                if ((pk == WP_LIN_P || pk == WP_LIN_U) && smol_wave_all_opaque (all_a))
                    put (WidePack<WP_LIN_O> (J, lut));
                else if (pk == WP_LIN_P)
                    put (WidePack<WP_LIN_P> (J, lut));
                else if (pk == WP_LIN_U)
                    put (WidePack<WP_LIN_U> (J, lut));
                else if (pk == WP_LIN_Q)
                    put (WidePack<WP_LIN_Q> (J, lut));
                else if (pk == WP_P16L)
                    put (WidePack<WP_P16L> (J, lut));
                else
                    put (WidePack<WP_ANY> (J, lut));
                s_y = 0;
                out_row += J.dst_stride;
            }
            else if (s_y == nsy)
            {
                uint32_t w [CPL];
#pragma unroll
                for (int c = 0; c < CPL; c++)
                {
                    const uint32_t a = ((acc [c][3] >> hv) >> 8) & 0xffu;
                    const uint32_t inv = inv_lut [a];
                    /* (s * inv) >> 16 & 0xff (:1036-1100): s < 2^16, inv <= 2^16, the product fits 32 bits */
                    const uint32_t cr = (__umul24 ((acc [c][0] >> hv) & 0xffffu, inv) >> 16) & 0xffu;
                    const uint32_t cg = (__umul24 ((acc [c][1] >> hv) & 0xffffu, inv) >> 16) & 0xffu;
                    const uint32_t cb = (__umul24 ((acc [c][2] >> hv) & 0xffffu, inv) >> 16) & 0xffu;
                    w [c] = (cr << dr) | (cg << dg) | (cb << db) | (a << da);
                }
                if (CPL == 4 && interior)
                {
                    const u32x4 v = { w [0], w [1 % CPL], w [2 % CPL], w [3 % CPL] };
                    /* non-temporal (this kernel is bound by arithmetic: the write-through asm store and its compiler
                     * barrier cost 66 -> 88 us on the premul16 ratio-1.5 case) */
                    store_row128 (out_row + (size_t) xl * 4, v, J.out_wt);
                }
                else
                {
#pragma unroll
                    for (int c = 0; c < CPL; c++)
                    {
                        if (xl + c < xn)
                        {
                            gbyte *p = out_row + (size_t) (xl + c) * 4;
                            if (J.out_dword_ok)
                                __builtin_nontemporal_store (w [c], reinterpret_cast<SMOL_AS_GLOBAL uint32_t *> (p));
                            else
                            {
                                p [0] = (uint8_t) w [c]; p [1] = (uint8_t) (w [c] >> 8);
                                p [2] = (uint8_t) (w [c] >> 16); p [3] = (uint8_t) (w [c] >> 24);
                            }
                        }
                    }
                }
                s_y = 0;
                out_row += J.dst_stride;
            }
            if (ks < n_samples)
            {
                e = (ks < 64u) ? smol_readlane (sched [0], ks) : smol_readlane (sched [1], ks - 64u);
                oy = e & 0xffffu;
                Fy = e >> 16;
            }
        }
#pragma unroll
        for (int c = 0; c < CPL; c++)
        {
            hp0 [c] = h0 [c];
            hp1 [c] = h1 [c];
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            hp32 [i] = h32 [i];

    }
}

/* table-driven flavours: 4 waves per workgroup share one LDS copy of the tables, each
 * wave works on its own tile in its own LDS region */
template <int HX, int CPL, int MODE>
__device__ __forceinline__ void down_gen_entry (const JobDev &J, uint8_t *lds)
{
#ifdef SMOL_KERNEL_EMU
    static uint8_t lut_lds [LUT_LDS_BYTES];
    lut_fill_lds (lut_lds, 0, 1);                       /* emulated: one wave per block */
    down_body<HX, CPL, MODE> (J, lds, lut_lds, threadIdx.x, blockIdx.x);
#else
    const uint32_t wave = smol_wave_index ();       /* a SCALAR: everything derived from the tile index stays wave-uniform */
    lut_fill_lds (lds, threadIdx.x, 256);
    __syncthreads ();
    const uint32_t pitch = (J.span_px * (MODE == DOWN_GEN32 ? 16u : 8u) + 15u) & ~15u;
    down_body<HX, CPL, MODE> (J, lds + ((LUT_LDS_BYTES + 15u) & ~15u) + wave * 2u * pitch, lds,
                              threadIdx.x & 63u, blockIdx.x * 4u + wave);
#endif
}

#ifdef SMOL_KERNEL_EMU
/* host emulation: one entry that switches at run time */
__global__ void k_down_one (const JobDev J)
{
    if (J.down_mode == 3)
    {
        const int in = J.p16 ? 2 : J.bpp_in == 3 ? 1 : 0;
        switch (J.halv_h * 3 + in)
        {
            case 0: lind_entry<0, false, false> (J, smol_dyn_lds); break;
            case 1: lind_entry<0, true, false> (J, smol_dyn_lds); break;
            case 2: lind_entry<0, false, true> (J, smol_dyn_lds); break;
            case 3: lind_entry<1, false, false> (J, smol_dyn_lds); break;
            case 4: lind_entry<1, true, false> (J, smol_dyn_lds); break;
            case 5: lind_entry<1, false, true> (J, smol_dyn_lds); break;
            case 6: lind_entry<2, false, false> (J, smol_dyn_lds); break;
            case 7: lind_entry<2, true, false> (J, smol_dyn_lds); break;
            default: lind_entry<2, false, true> (J, smol_dyn_lds); break;
        }
        return;
    }
    if (J.down_mode == DOWN_GEN16 || J.down_mode == DOWN_GEN32)
    {
        const bool w32 = J.down_mode == DOWN_GEN32;
        if (J.tile_cpl == 4)
        {
            down_gen_entry<0, 4, DOWN_GEN16> (J, smol_dyn_lds);
            return;
        }
        switch (J.halv_h)
        {
            case 0: if (w32) down_gen_entry<0, 1, DOWN_GEN32> (J, smol_dyn_lds); else down_gen_entry<0, 1, DOWN_GEN16> (J, smol_dyn_lds); break;
            case 1: if (w32) down_gen_entry<1, 1, DOWN_GEN32> (J, smol_dyn_lds); else down_gen_entry<1, 1, DOWN_GEN16> (J, smol_dyn_lds); break;
            default: if (w32) down_gen_entry<2, 1, DOWN_GEN32> (J, smol_dyn_lds); else down_gen_entry<2, 1, DOWN_GEN16> (J, smol_dyn_lds); break;
        }
        return;
    }
    if (J.tile_cpl == 4)
    {
        down_body<0, 4, DOWN_P16> (J, smol_dyn_lds, nullptr, threadIdx.x, blockIdx.x);
        return;
    }
    switch (J.halv_h)
    {
        case 0: down_body<0, 1, DOWN_P16> (J, smol_dyn_lds, nullptr, threadIdx.x, blockIdx.x); break;
        case 1: down_body<1, 1, DOWN_P16> (J, smol_dyn_lds, nullptr, threadIdx.x, blockIdx.x); break;
        default: down_body<2, 1, DOWN_P16> (J, smol_dyn_lds, nullptr, threadIdx.x, blockIdx.x); break;
    }
}
#else
/* (amdgpu_waves_per_eu 6 or 8 on these kernels: premul16 ratio 1.5 57 -> 85 / 147 us -- spills --, the table-driven
 * flavours within 3 % either way: profiles/r03i_down_occ_ab.txt) */
/* HC: horizontal halvings 0-2 with one column per lane, or 3 = no halvings, 4 columns per lane */
template <int HC>
__global__ void __launch_bounds__ (64)
k_down_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.y);
    down_body<HC == 3 ? 0 : HC, HC == 3 ? 4 : 1, DOWN_P16> (J, smol_dyn_lds, nullptr, threadIdx.x, blockIdx.x);
}

template <int HC>
__global__ void __launch_bounds__ (64)
k_down_one (const JobDev J)
{
    down_body<HC == 3 ? 0 : HC, HC == 3 ? 4 : 1, DOWN_P16> (J, smol_dyn_lds, nullptr, threadIdx.x, blockIdx.x);
}

/* HC: horizontal halvings 0-2 with one column per lane, or 3 = no halvings, 4 columns per lane (DOWN_GEN16 only) */
template <int HC, int MODE>
__global__ void __launch_bounds__ (256)
k_downg_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.y);
    down_gen_entry<HC == 3 ? 0 : HC, HC == 3 ? 4 : 1, MODE> (J, smol_dyn_lds);
}

template <int HC, int MODE>
__global__ void __launch_bounds__ (256)
k_downg_one (const JobDev J)
{
    down_gen_entry<HC == 3 ? 0 : HC, HC == 3 ? 4 : 1, MODE> (J, smol_dyn_lds);
}
#endif

} /* namespace smolhip */
