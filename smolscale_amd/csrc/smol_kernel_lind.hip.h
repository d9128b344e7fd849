/* smol_kernel_lind.hip.h -- k_lind: bilinear MINIFICATION on both axes in the 128bpp class, 0-2 halvings each:
 * gamma-correct (sRGB-linear, 11-bit channels; the with_srgb photo / PNG downscale), premul16 and premul16 + sRGB
 * (32-bit channels) (unpack smolscale-generic.c:481-535, 588-614; horizontal :1290-1425; vertical :1648-2007; pack
 * :922-935, 1010-1023, 1096-1134). Round 4's successor of k_down and k_downg, which it replaced on every row; its three
 * row passes also serve the mixed-axis roads (k_linh, k_linv below).
 *
 * What the old kernel's counters said (profiles/r03final_sq_srgbhalf.txt): 102 vector lane-operations per SOURCE pixel
 * where the arithmetic needs about 50, and 0.59 scalar instructions per vector one. The lanes were there, idle: a tile
 * of 64 output columns spans 97 source pixels at ratio 1.5, converted in three rounds of 64 (51 % of the slots used),
 * every round and every store behind its own exec-mask branch. Here:
 *
 *   tile    W output columns x tile_rows rows per wave, W <= 128 chosen on the HOST (lind_setup) so that the tile's span
 *           of source pixels just fills K whole rounds of 64 (ratio 1.5: W = 126, span 191 of 192 slots). A lane owns
 *           columns x0 + lane and x0 + 64 + lane: a wave's stores and LDS reads are runs of neighbouring addresses.
 *   stage   per source row and round: one raw pixel per lane (index clamped into the span, so that every slot holds a
 *           real pixel and nothing needs a mask), converted ONCE (WideUnpack, flavour fixed per row: opaque rows take
 *           the fused table) and written to the wave's single LDS row buffer as four 16-bit channels. Row r + 1 is in
 *           flight in registers while row r is filtered. A 24-bit pixel is ONE dword load at its own byte address
 *           (global loads need no alignment on gfx950; the row's last pixel is read one byte early and shifted, so
 *           that nothing past the row is touched).
 *   H, V    packed 16-bit taps (lerp11_pk), the vertical schedule read back with v_readlane.
 *   pack    WidePack, flavour fixed per output row (opaque rows skip the inverse-alpha look-up).
 *
 * No branch depends on a lane: the only masks are on the stores of a tile's partial last columns. */

#pragma once

namespace smolhip {

constexpr int LIND_KMAX = 5;            /* conversion rounds per source row: spans of at most 320 pixels */
constexpr int LIND_KMAX_NARROW = 8;     /* the one-column flavour (NARROW: half the registers per lane): 512 pixels, 64 columns up to a ratio of 8 */
typedef uint32_t u32a1 __attribute__ ((aligned (1)));

/* one tap on a 32-bit channel of at most 19 bits: (p * F + q * (256 - F)) >> 8, below 2^28 before the shift */
__device__ __forceinline__ uint32_t tap32 (uint32_t p, uint32_t q, uint32_t F, uint32_t G)
{
    return ((uint32_t) __umul24 (p, F) + (uint32_t) __umul24 (q, G)) >> 8;
}

/* W32: premul16 + sRGB-linear (unassociated in AND out with with_srgb: 19-bit channels, :627-660 and :1135-1158):
 * four 32-bit channels per staged pixel, taps as 24-bit multiplies (p * F + q * (256 - F)) >> 8 */
/* PASS: the two halves of the kernel as passes of their own through the intermediate image of the two-pass road
 * (smol_jobdev.h: twopass_setup), for gamma-correct jobs whose axes differ:
 *   LIND_FUSED  both axes here (minification on both)
 *   LIND_H      k_linh: staging + horizontal filter of `tile_rows' SOURCE rows, result (two packed channel pairs, 8 bytes)
 *               to J.tmp; any horizontal step (a magnifying axis spans few source pixels per tile)
 *   LIND_V      k_linv: rows of J.tmp (from k_linh or k_boxh) through the vertical filter and the packer; nothing staged */
enum : int { LIND_FUSED = 0, LIND_H = 1, LIND_V = 2 };

/* TMP8 (passes of the W32 flavour only): the intermediate pixel is 8 bytes of 16-bit channels -- what k_boxh writes and
 * k_boxv reads (premul16 without sRGB: every value fits 16 bits) -- instead of four 32-bit channels (16 bytes) */
/* NARROW (chosen per job at the entry, one or two halvings): tiles of at most 64 columns */
template <int HX, bool IN24, bool W32, int PASS = LIND_FUSED, bool TMP8 = false, bool NARROW = false>
__device__ __forceinline__ void lind_body (const JobDev &J, uint8_t *buf, const uint8_t *lut_lds,
                                           const uint32_t lane, const uint32_t tile)
{
    static_assert (!(IN24 && W32), "premul16 means 32-bit pixels on both sides");
    constexpr uint32_t TPX = (W32 && !TMP8) ? 16u : 8u;            /* bytes per pixel of the intermediate image */
    constexpr uint32_t NSX = 1u << HX;
    constexpr int CPL = 2;
    constexpr uint32_t PXB = W32 ? 16u : 8u;                        /* LDS bytes per staged pixel */
    constexpr int NCH = W32 ? 4 : 2;                                /* registers per filtered pixel */
    const uint32_t W = PASS == LIND_V ? 128u : J.group_rows;        /* output columns per tile (host) */
    /* tiles of at most 64 columns (minification beyond ~5: the span of 64 columns fills the conversion rounds) leave the
     * lanes' second column empty: nothing is filtered, packed or stored for it -- a compile-time flavour, picked per job
     * (as a run-time test it cost the wide tiles 1-3 %): gamma-correct ratio 6 125 -> 104 us, photographs at 7.1 151 -> 128 */
    constexpr int ncols = NARROW ? 1 : CPL;
    constexpr int KMAX = NARROW ? LIND_KMAX_NARROW : LIND_KMAX;

    const uint32_t tiles_x = (J.w_out + W - 1) / W;
    const uint32_t tx = tile % tiles_x;
    const uint32_t ty = tile / tiles_x;
    const uint32_t yrel0 = ty * J.tile_rows;                        /* LIND_H: counted in rows of the intermediate image */

    if (PASS == LIND_H ? J.tmp_row0 + yrel0 > J.tmp_row1 : yrel0 >= J.n_rows)
        return;

    const uint32_t yrel1 = (yrel0 + J.tile_rows < J.n_rows) ? yrel0 + J.tile_rows : J.n_rows;
    const uint32_t x0 = tx * W;
    const uint32_t xn = (x0 + W < J.w_out) ? x0 + W : J.w_out;

    /* ---- horizontal samples of this lane's two columns ---------------------------- */
    uint32_t xc [CPL];
    bool ok [CPL];
    uint32_t tap_ofs [CPL][NSX], tap_f [CPL][NSX];
    Lerp11W tap_w [CPL][NSX];
#pragma unroll
    for (int c = 0; c < CPL; c++)
    {
        xc [c] = x0 + 64u * c + lane;
        ok [c] = xc [c] < xn;
        const uint32_t x = ok [c] ? xc [c] : xn - 1;
        if (PASS == LIND_V)
            xc [c] = x;                                             /* (the column whose intermediate pixels the lane reads) */
        uint64_t pos = J.frac0_x + (uint64_t) (x << HX) * J.step_x;
#pragma unroll
        for (uint32_t s = 0; s < NSX; s++)
        {
            bilin_tap (pos, J.w_in, tap_ofs [c][s], tap_f [c][s]);
            if (PASS != LIND_V && J.filter_h == F_ONE)
            {
                /* an axis of ONE pixel (:1563-1611): every column is pixel 0 -- weight 256 on it, and the neighbour the
                 * tap also reads is the clamped copy of it in the next slot */
                tap_ofs [c][s] = 0;
                tap_f [c][s] = 256;
            }
            tap_w [c][s] = lerp11_weights (tap_f [c][s]);
            pos += J.step_x;
        }
    }
    const uint32_t sx0 = PASS == LIND_V ? 0u : smol_readlane (tap_ofs [0][0], 0);     /* first source pixel of the tile */
    uint32_t sx1, f_unused;
    bilin_tap (J.frac0_x + (uint64_t) ((xn << HX) - 1) * J.step_x, J.w_in, sx1, f_unused);
    sx1 += 1;                                                       /* last source pixel of the tile  */
    if (PASS != LIND_V && J.filter_h == F_ONE)
        sx1 = 0;
    const uint32_t npx = PASS == LIND_V ? 0u : sx1 - sx0 + 1;       /* pixels per staged row: <= 64 * LIND_KMAX (host) */
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (uint32_t s = 0; s < NSX; s++)
            tap_ofs [c][s] = (tap_ofs [c][s] - sx0) * PXB;          /* byte offset in the LDS row */

    /* ---- vertical schedule: sample k of the tile in lane k & 63 of sched [k >> 6] --- */
    const uint32_t hv = J.halv_v;
    const uint32_t nsy = 1u << hv;
    const uint32_t jy0 = (J.first_row + yrel0) << hv;
    const uint32_t n_samples = PASS == LIND_H ? 1u : (yrel1 - yrel0) << hv;     /* <= 128 (lind_setup) */
    uint32_t sched [2] = { 0, 0 };
    uint32_t e = 0, e_last = 0;
    if (PASS != LIND_H)
    {
#pragma unroll
        for (uint32_t q = 0; q < 2; q++)
        {
            uint32_t k = q * 64u + lane;
            if (k >= n_samples)
                k = n_samples - 1;
            uint32_t o, f;
            bilin_tap (J.frac0_y + (uint64_t) (jy0 + k) * J.step_y, J.h_in, o, f);
            if (PASS != LIND_H && (J.filter_v == F_COPY || J.filter_v == F_ONE))
            {
                /* a COPY axis: output row R is intermediate row R and nothing else -- the strip's slab of source rows
                 * holds no neighbour to give weight 0 to (smol_hip_input_row_range); an axis of ONE row: always row 0 */
                o = J.filter_v == F_ONE ? 0u : jy0 + k;
                f = 0;
            }
            sched [q] = o | (f << 16);
        }
        e = smol_readlane (sched [0], 0);
        e_last = smol_readlane (sched [(n_samples - 1) >> 6], (n_samples - 1) & 63u);
    }
    /* a sample is due when its BOTTOM row has passed: row o + 1 of a bilinear pair, row o itself on a COPY axis */
    const uint32_t vdue = (PASS != LIND_H && (J.filter_v == F_COPY || J.filter_v == F_ONE)) ? 0u : 1u;
    uint32_t oy = e & 0xffffu, Fy = e >> 16;
    const uint32_t row_first = PASS == LIND_H ? J.tmp_row0 + yrel0 : oy;
    const uint32_t row_last = PASS == LIND_H ? (row_first + J.tile_rows - 1u < J.tmp_row1 ? row_first + J.tile_rows - 1u : J.tmp_row1)
                                             : (e_last & 0xffffu) + vdue;

    /* ---- staging -------------------------------------------------------------------- */
    const LutLds lut (lut_lds);
    const int uk = wide_unpack_kind (J), pk = wide_pack_kind (J);
    const gbyte *srow = gptr (J.src) + (size_t) row_first * J.src_stride;   /* row being loaded */
    /* slot k * 64 + lane holds pixel min (slot, npx - 1) of the span; px_off: its byte offset in a source row */
    uint32_t px_off [KMAX], px_sh [IN24 ? KMAX : 1];
#pragma unroll
    for (int k = 0; k < KMAX; k++)
    {
        const uint32_t i = (uint32_t) k * 64u + lane;
        const uint32_t x = sx0 + (i < npx ? i : npx - 1);
        if (IN24)
        {
            /* the dword at the row's last pixel would end one byte past the row: read it one byte early */
            px_sh [k] = x == J.w_in - 1u ? 1u : 0u;
            px_off [k] = x * 3u - px_sh [k];
        }
        else
            px_off [k] = x * 4u;
    }
    uint32_t rawpx [KMAX];                                     /* the pixels of the row in flight */
    /* LIND_V: the lane's two intermediate pixels of the row in flight */
    uint32_t tmp_in [CPL][NCH];
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int i = 0; i < NCH; i++)
            tmp_in [c][i] = 0;
    /* LIND_H: row being written; LIND_V: row being loaded (the fused kernel has no intermediate image) */
    const gbyte *trow = PASS == LIND_FUSED ? nullptr : gptr (J.tmp) + (size_t) (row_first - J.tmp_row0) * J.w_out * TPX;

    auto load_row = [&] ()
    {
        if (PASS == LIND_V)
        {
#pragma unroll
            for (int c = 0; c < CPL; c++)
            {
                if (TPX == 16)
                {
                    const u32x4 t = *reinterpret_cast<const SMOL_AS_GLOBAL u32x4 *> (trow + (size_t) xc [c] * 16u);
                    tmp_in [c][0] = t.x; tmp_in [c][1] = t.y; tmp_in [c][NCH - 2] = t.z; tmp_in [c][NCH - 1] = t.w;
                }
                else
                {
                    const u32x2a8 t = *reinterpret_cast<const SMOL_AS_GLOBAL u32x2a8 *> (trow + (size_t) xc [c] * 8u);
                    if (W32)
                    {
                        tmp_in [c][0] = t.x & 0xffffu; tmp_in [c][1] = t.x >> 16;
                        tmp_in [c][NCH - 2] = t.y & 0xffffu; tmp_in [c][NCH - 1] = t.y >> 16;
                    }
                    else
                    {
                        tmp_in [c][0] = t.x;
                        tmp_in [c][NCH - 1] = t.y;
                    }
                }
            }
            trow += (size_t) J.w_out * TPX;
            return;
        }
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if ((uint32_t) k * 64u < npx)                           /* wave-uniform */
            {
                if (IN24)
                {
#ifdef SMOL_KERNEL_EMU
                    __builtin_memcpy (&rawpx [k], srow + px_off [k], 4);
#else
                    rawpx [k] = *reinterpret_cast<const SMOL_AS_GLOBAL u32a1 *> (srow + px_off [k]);
#endif
                }
                else if (J.in_dword_ok)
                    rawpx [k] = *reinterpret_cast<const SMOL_AS_GLOBAL uint32_t *> (srow + px_off [k]);
                else
                {
                    /* 32-bit pixels at any byte address (wave-uniform choice): the same unaligned dword load */
#ifdef SMOL_KERNEL_EMU
                    __builtin_memcpy (&rawpx [k], srow + px_off [k], 4);
#else
                    rawpx [k] = *reinterpret_cast<const SMOL_AS_GLOBAL u32a1 *> (srow + px_off [k]);
#endif
                }
            }
        srow += J.src_stride;
    };

    auto convert_row = [&] ()
    {
        if (IN24)
        {
#pragma unroll
            for (int k = 0; k < KMAX; k++)
                if ((uint32_t) k * 64u < npx)
                    rawpx [k] = __builtin_amdgcn_alignbyte (0u, rawpx [k], px_sh [k]);
        }
        auto conv = [&] (const auto &un)
        {
#pragma unroll
            for (int k = 0; k < KMAX; k++)
                if ((uint32_t) k * 64u < npx)
                {
                    uint32_t ch [4];
                    un (rawpx [k], ch);
                    if (W32)
                        *reinterpret_cast<u32x4 *> (buf + ((uint32_t) k * 64u + lane) * 16u) = u32x4 { ch [0], ch [1], ch [2], ch [3] };
                    else
                        *reinterpret_cast<u32x2a8 *> (buf + ((uint32_t) k * 64u + lane) * 8u)
                            = u32x2a8 { ch [0] | (ch [1] << 16), ch [2] | (ch [3] << 16) };
                }
        };
        if (W32)
        {
            if (J.linear)
                conv (WideUnpack<WU_P16L> (J, lut));
            else
                conv (WideUnpack<WU_P16> (J, lut));
        }
        else if (PASS == LIND_H && !J.linear)
            conv (WideUnpack<WU_ANY> (J, lut));         /* the class's compressed storage (a box of 256+ pixels on the other axis): bytes as they are */
        else if (IN24 || uk == WU_LIN_O)
            conv (WideUnpack<WU_LIN_O> (J, lut));
        else
        {
            uint32_t all = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < KMAX; k++)
                if ((uint32_t) k * 64u < npx)
                    all &= rawpx [k];
            if (wide_batch_opaque (J, all))
                conv (WideUnpack<WU_LIN_O> (J, lut));
            else if (uk == WU_LIN_P)
                conv (WideUnpack<WU_LIN_P> (J, lut));
            else
                conv (WideUnpack<WU_LIN_U> (J, lut));
        }
    };

    /* ---- the row loop ----------------------------------------------------------------- */
    gbyte *out_row = gptr (J.dst) + (size_t) yrel0 * J.dst_stride;
    uint32_t ks = 0, s_y = 0;
    /* previous row's horizontal result and the vertical sums: two packed channel pairs (R | G << 16, B | A << 16),
     * or four 32-bit channels (W32) */
    uint32_t hp [CPL][NCH], acc [CPL][NCH];
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int i = 0; i < NCH; i++)
            hp [c][i] = acc [c][i] = 0;

    load_row ();

    for (uint32_t r = row_first; r <= row_last; r++)
    {
        uint32_t h [CPL][NCH];
        if (PASS == LIND_V)
        {
#pragma unroll
            for (int c = 0; c < CPL; c++)
#pragma unroll
                for (int i = 0; i < NCH; i++)
                    h [c][i] = tmp_in [c][i];
        }
        else
        {
            smol_wave_sync ();                          /* row r - 1 has been read (a real barrier only in the host emulation) */
            convert_row ();                             /* waits for row r's loads */
            smol_wave_sync ();
        }
        if (r < row_last)
            load_row ();                                /* row r + 1 flies while row r is filtered */

        /* ---- H ---------------------------------------------------------------------------------- */
#pragma unroll
        for (int c = 0; c < CPL && PASS != LIND_V; c++)
        {
            if (c >= ncols)
                continue;
            uint32_t hs [NCH];
#pragma unroll
            for (uint32_t s = 0; s < NSX; s++)
            {
                if (W32)
                {
                    const u32x4 pp = *reinterpret_cast<const u32x4 *> (buf + tap_ofs [c][s]);
                    const u32x4 qq = *reinterpret_cast<const u32x4 *> (buf + tap_ofs [c][s] + 16);
                    const uint32_t F = tap_f [c][s], G = 256u - F;
                    const uint32_t t [4] = { tap32 (pp.x, qq.x, F, G), tap32 (pp.y, qq.y, F, G), tap32 (pp.z, qq.z, F, G), tap32 (pp.w, qq.w, F, G) };
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        hs [i] = s ? hs [i] + t [i] : t [i];
                }
                else
                {
                    /* both channel pairs stay packed; the sums of 2^HX taps stay below 2^13 per lane */
                    const u32x4a8 pq = *reinterpret_cast<const u32x4a8 *> (buf + tap_ofs [c][s]);
                    const uint32_t t01 = lerp11_pk (pq.x, pq.z, tap_w [c][s]);
                    const uint32_t t23 = lerp11_pk (pq.y, pq.w, tap_w [c][s]);
                    hs [0] = s ? hs [0] + t01 : t01;
                    hs [1] = s ? hs [1] + t23 : t23;
                }
            }
#pragma unroll
            for (int i = 0; i < NCH; i++)
                h [c][i] = !HX ? hs [i] : W32 ? hs [i] >> HX : (hs [i] >> HX) & (0xffffu >> HX) * 0x00010001u;
        }

        if (PASS == LIND_H)
        {
            /* the horizontal result of this source row: row r of the intermediate image */
#pragma unroll
            for (int c = 0; c < CPL; c++)
                if (ok [c])
                {
                    gbyte *tp = const_cast<gbyte *> (trow) + (size_t) xc [c] * TPX;
                    if (TPX == 16)
                        *reinterpret_cast<SMOL_AS_GLOBAL u32x4 *> (tp) = u32x4 { h [c][0], h [c][1], h [c][NCH - 2], h [c][NCH - 1] };
                    else if (W32)
                        *reinterpret_cast<SMOL_AS_GLOBAL u32x2a8 *> (tp) = u32x2a8 { h [c][0] | (h [c][1] << 16), h [c][NCH - 2] | (h [c][NCH - 1] << 16) };
                    else
                        *reinterpret_cast<SMOL_AS_GLOBAL u32x2a8 *> (tp) = u32x2a8 { h [c][0], h [c][NCH - 1] };
                }
            trow += (size_t) J.w_out * TPX;
            continue;
        }

        /* ---- V: every sample whose bottom row is r ---------------------------------------------- */
        while (ks < n_samples && oy + vdue == r)
        {
            const Lerp11W wy = lerp11_weights (Fy);
            const uint32_t Gy = 256u - Fy;
#pragma unroll
            for (int c = 0; c < CPL; c++)
                if (c < ncols)
                {
#pragma unroll
                    for (int i = 0; i < NCH; i++)
                    {
                        const uint32_t t = W32 ? tap32 (hp [c][i], h [c][i], Fy, Gy) : lerp11_pk (hp [c][i], h [c][i], wy);
                        acc [c][i] = s_y ? acc [c][i] + t : t;
                    }
                }
            s_y++;
            ks++;
            if (s_y == nsy)
            {
                uint32_t v [CPL][4], all_a = 255u;
#pragma unroll
                for (int c = 0; c < CPL; c++)
                {
                    if (c >= ncols)
                        continue;
                    if (W32)
                    {
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            v [c][i] = acc [c][i] >> hv;
                    }
                    else
                    {
                        v [c][0] = (acc [c][0] & 0xffffu) >> hv; v [c][1] = acc [c][0] >> (16u + hv);
                        v [c][2] = (acc [c][1] & 0xffffu) >> hv; v [c][3] = acc [c][1] >> (16u + hv);
                        all_a &= v [c][3] & 0xffu;      /* (columns past the tile's end repeat its last one) */
                    }
                }
                auto put = [&] (const auto &pack)
                {
#pragma unroll
                    for (int c = 0; c < CPL; c++)
                    {
                        if (c >= ncols)
                            continue;
                        const uint32_t w = pack (v [c]);
                        if (ok [c])
                        {
                            if (J.bpp_out == 4 && J.out_dword_ok)
                                *reinterpret_cast<SMOL_AS_GLOBAL uint32_t *> (out_row + (size_t) xc [c] * 4) = w;     /* (plain: the kernel is bound by its arithmetic) */
                            else
                                wide_store_word (J, out_row, xc [c], w);
                        }
                    }
                };
                /* once per output row and wave; all-opaque rows skip the inverse-alpha look-up and multiplies */
                if (W32)
                {
                    if (J.linear)
                        put (WidePack<WP_P16L> (J, lut));
                    else
                        put (WidePack<WP_P16> (J, lut));
                }
                else if (PASS == LIND_V && pk == WP_ANY)
                    put (WidePack<WP_ANY> (J, lut));     /* compressed storage behind a box of 256+ pixels on the other axis */
                else if ((pk == WP_LIN_P || pk == WP_LIN_U) && smol_wave_all_opaque (all_a))
                    put (WidePack<WP_LIN_O> (J, lut));
                else if (pk == WP_LIN_P)
                    put (WidePack<WP_LIN_P> (J, lut));
                else if (pk == WP_LIN_U)
                    put (WidePack<WP_LIN_U> (J, lut));
                else
                    put (WidePack<WP_LIN_Q> (J, lut));
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
            if (c < ncols)
            {
#pragma unroll
                for (int i = 0; i < NCH; i++)
                    hp [c][i] = h [c][i];
            }
    }
}

/* LDS of one wave: the row buffer */
__device__ __forceinline__ uint32_t lind_wave_lds (bool w32, bool narrow = false)
{
    return 64u * (narrow ? LIND_KMAX_NARROW : LIND_KMAX) * (w32 ? 16u : 8u);
}

/* 4 waves per workgroup share one LDS copy of the tables, each wave works on its own tile in its own LDS region */
/* NARROW: the one-column flavour, a kernel of its own (k_lind_batch<HX, IN, true>; as a second body inside the common
 * kernel it cost the wide tiles 3-4 %, as k_boxh's 64-lane flavour did); the host picks it per job: dispatch::lind_narrow */
template <int HX, bool IN24, bool W32, bool NARROW = false>
__device__ __forceinline__ void lind_entry (const JobDev &J, uint8_t *lds)
{
#ifdef SMOL_KERNEL_EMU
    static uint8_t lut_lds [LUT_LDS_BYTES];
    lut_fill_lds (lut_lds, 0, 1);                       /* emulated: one wave per block */
    lind_body<HX, IN24, W32, LIND_FUSED, false, NARROW> (J, lds, lut_lds, threadIdx.x, blockIdx.x);
#else
    /* (readfirstlane: the compiler cannot know that threadIdx.x >> 6 is the same on every lane of a wave, and
     * without it everything derived from the tile index -- every loop bound, every "is this round needed" test --
     * is computed per lane and branched on with exec masks) */
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane ((int) (threadIdx.x >> 6));
    lut_fill_lds (lds, threadIdx.x, 256);
    __syncthreads ();
    uint8_t *tab_end = lds + ((LUT_LDS_BYTES + 15u) & ~15u);       /* (the host sizes the workgroup's LDS by the flavour too: down_lds_bytes) */
    lind_body<HX, IN24, W32, LIND_FUSED, false, NARROW> (J, tab_end + wave * lind_wave_lds (W32, NARROW), lds, threadIdx.x & 63u, blockIdx.x * 4u + wave);
#endif
}

/* k_linh: blockIdx.x = column tile, the 4 waves of a workgroup take 4 row tiles one below the other (smol_jobdev.h:
 * box2_h_grid); k_linv: 4 tiles per workgroup, numbered row by row.
 * KIND: 0 = 11-bit channels, 32-bit rows; 1 = 11-bit channels, 24-bit rows; 2 = premul16 + sRGB between two k_lin* passes
 * (19-bit channels, 16-byte intermediate pixels); 3 = every other premul16 job (8-byte intermediate pixels) */
template <int HX, int KIND>
__device__ __forceinline__ void linh_entry (const JobDev &J, uint8_t *lds)
{
#ifdef SMOL_KERNEL_EMU
    static uint8_t lut_lds [LUT_LDS_BYTES];
    lut_fill_lds (lut_lds, 0, 1);
    lind_body<HX, KIND == 1, KIND >= 2, LIND_H, KIND == 3> (J, lds, lut_lds, threadIdx.x, blockIdx.x);     /* emulated: blocks numbered tile by tile */
#else
    const uint32_t wave = smol_wave_index ();
    lut_fill_lds (lds, threadIdx.x, 256);
    __syncthreads ();
    const uint32_t tiles_x = (J.w_out + J.group_rows - 1u) / J.group_rows;
    lind_body<HX, KIND == 1, KIND >= 2, LIND_H, KIND == 3> (J, lds + ((LUT_LDS_BYTES + 15u) & ~15u) + wave * lind_wave_lds (KIND >= 2), lds,
                                                            threadIdx.x & 63u, (blockIdx.y * 4u + wave) * tiles_x + blockIdx.x);
#endif
}

template <int KIND>
__device__ __forceinline__ void linv_entry (const JobDev &J, uint8_t *lds)
{
#ifdef SMOL_KERNEL_EMU
    static uint8_t lut_lds [LUT_LDS_BYTES];
    lut_fill_lds (lut_lds, 0, 1);
    lind_body<0, false, KIND >= 2, LIND_V, KIND == 3> (J, lds, lut_lds, threadIdx.x, blockIdx.x);
#else
    lut_fill_lds (lds, threadIdx.x, 256);
    __syncthreads ();
    lind_body<0, false, KIND >= 2, LIND_V, KIND == 3> (J, lds, lds, threadIdx.x & 63u, blockIdx.x * 4u + smol_wave_index ());
#endif
}

/* the KIND of a two-pass job (host twin: smol_k_box2.hip) */
__device__ __forceinline__ int lin_pass_kind (const JobDev &J)
{
    if (!J.p16)
        return J.bpp_in == 3 ? 1 : 0;
    /* premul16 without sRGB: every value fits 16 bits, whatever the passes; with sRGB (19 bits) only next to a box pass
     * (whose results the reference masks to 16 bits: smolscale-generic.c:1231-1261) */
    return (!J.linear || J.filter_h == F_BOX || J.filter_v == F_BOX) ? 3 : 2;
}

/* pass 1 / pass 2 of the two-pass road for a job's filters (k_boxh / k_boxv stay with their own entries) */
__device__ __forceinline__ void linh_dispatch (const JobDev &J, uint8_t *lds)
{
    switch (J.halv_h * 4 + lin_pass_kind (J))
    {
        case 0: linh_entry<0, 0> (J, lds); break;
        case 1: linh_entry<0, 1> (J, lds); break;
        case 2: linh_entry<0, 2> (J, lds); break;
        case 3: linh_entry<0, 3> (J, lds); break;
        case 4: linh_entry<1, 0> (J, lds); break;
        case 5: linh_entry<1, 1> (J, lds); break;
        case 6: linh_entry<1, 2> (J, lds); break;
        case 7: linh_entry<1, 3> (J, lds); break;
        case 8: linh_entry<2, 0> (J, lds); break;
        case 9: linh_entry<2, 1> (J, lds); break;
        case 10: linh_entry<2, 2> (J, lds); break;
        default: linh_entry<2, 3> (J, lds); break;
    }
}

__device__ __forceinline__ void linv_dispatch (const JobDev &J, uint8_t *lds)
{
    switch (lin_pass_kind (J))
    {
        case 2: linv_entry<2> (J, lds); break;
        case 3: linv_entry<3> (J, lds); break;
        default: linv_entry<0> (J, lds); break;
    }
}

#ifndef SMOL_KERNEL_EMU
/* IN: 0 = 32-bit rows, 11-bit channels; 1 = 24-bit rows; 2 = 32-bit rows, premul16 + linear (19-bit channels) */
template <int HX, int IN, bool NARROW = false>
__global__ void __launch_bounds__ (256)
k_lind_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.y);
    lind_entry<HX, IN == 1, IN == 2, NARROW> (J, smol_dyn_lds);
}

template <int HX, int IN, bool NARROW = false>
__global__ void __launch_bounds__ (256)
k_lind_one (const JobDev J)
{
    lind_entry<HX, IN == 1, IN == 2, NARROW> (J, smol_dyn_lds);
}

/* (pass 1: jobs along blockIdx.z like k_boxh; pass 2: along blockIdx.y like k_boxv) */
template <int HX, int KIND>
__global__ void __launch_bounds__ (256)
k_linh_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.z);
    linh_entry<HX, KIND> (J, smol_dyn_lds);
}

template <int HX, int KIND>
__global__ void __launch_bounds__ (256)
k_linh_one (const JobDev J)
{
    linh_entry<HX, KIND> (J, smol_dyn_lds);
}

template <int KIND>
__global__ void __launch_bounds__ (256)
k_linv_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.y);
    linv_entry<KIND> (J, smol_dyn_lds);
}

template <int KIND>
__global__ void __launch_bounds__ (256)
k_linv_one (const JobDev J)
{
    linv_entry<KIND> (J, smol_dyn_lds);
}
#endif

} /* namespace smolhip */
