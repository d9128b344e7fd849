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
#if defined (SMOL_TUNING) && !defined (SMOL_KERNEL_EMU)
/* timing ablations, -DSMOL_TUNING builds only (env SMOL_LIND_DBG, set by the launcher): 1 no taps, 2 no loads, 4 no stores,
 * 8 source table look-ups replaced by shifts, 16 output table look-ups replaced by shifts, 32 no conversion at all */
static __device__ uint32_t smol_lind_dbg_dev;    /* (one per translation unit; the fused kernels live in smol_k_updown.hip) */
#define SMOL_LIND_DBG() (smol_lind_dbg_dev)
#else
#define SMOL_LIND_DBG() 0u
#endif

template <int HX, bool IN24, bool W32, int PASS = LIND_FUSED, bool TMP8 = false, bool NARROW = false, bool NT = false, bool HV0 = false>
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

    const uint32_t dbg = PASS == LIND_FUSED ? SMOL_LIND_DBG () : 0u;   /* (constant 0 in the shipped library) */
    const uint32_t tiles_x = (J.w_out + W - 1) / W;
    const uint32_t tx = tile % tiles_x;
    const uint32_t ty = tile / tiles_x;
    const uint32_t yrel0 = ty * J.tile_rows;                        /* LIND_H: counted in rows of the intermediate image */

    /* LIND_H under a tall vertical box (JobDev::box2 == 3, smol_jobdev.h): the tile is (column tile, output row, split);
     * it walks its share of the output row's span of source rows and keeps the box sums in registers */
    const bool hbox = PASS == LIND_H && dispatch::twopass_hbox (J);
    uint32_t hb_row = 0, hb_first = 0, hb_last = 0, hb_y0 = 0, hb_y1 = 0, hb_w1 = 0, hb_lastw = 0;
    bool hb_empty = false;
    if (hbox)
    {
        const uint32_t S = dispatch::hbox_splits (J);
        const uint32_t yo = ty / S, sp = ty - yo * S;
        if (yo >= J.n_rows)
            return;
        /* the vertical box of boxv_body<true> (smolscale-generic.c:2163-2260): first row weighted w1, last row w2 - 1
         * (skipped when w2 == 0), interior rows in full */
        uint32_t w2;
        box_span (J.step_y, J.h_in, J.first_row + yo, hb_y0, hb_y1, w2, hb_w1);
        hb_lastw = w2 ? w2 - 1u : 0u;
        const uint32_t y_end = w2 == 0u ? hb_y1 - 1u : hb_y1;
        const uint32_t chunk = (y_end - hb_y0 + S) / S;
        hb_first = hb_y0 + sp * chunk;
        hb_last = hb_first + chunk - 1u < y_end ? hb_first + chunk - 1u : y_end;
        hb_empty = hb_first > y_end;
        hb_row = ty;                                                /* row of the partial-sum image */
    }
    else if (PASS == LIND_H ? J.tmp_row0 + yrel0 > J.tmp_row1 : yrel0 >= J.n_rows)
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
    constexpr bool nt_out = NT;                                     /* (streaming stores: a kernel of its own, see lind3_body) */
    /* HV0 (and NT, a magnification on both axes): no vertical halving, ONE sample per output row, known at compile time -- no
     * shifts by zero, no sums of one term (gamma-correct x2 331 -> 310 us, 24-bit rows x4 158 -> 137: profiles/r05_lind_hv0.txt).
     * The launcher picks the flavour by the job's halv_v. */
    const uint32_t hv = (NT || HV0) ? 0u : J.halv_v;
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
    const uint32_t row_first = hbox ? hb_first : PASS == LIND_H ? J.tmp_row0 + yrel0 : oy;
    const uint32_t row_last = hbox ? hb_last
                            : PASS == LIND_H ? (row_first + J.tile_rows - 1u < J.tmp_row1 ? row_first + J.tile_rows - 1u : J.tmp_row1)
                                             : (e_last & 0xffffu) + vdue;

    /* ---- staging -------------------------------------------------------------------- */
    const LutLds lut (lut_lds, LUT_AT_LDS_0);       /* (the entries see to it: lut_expect_lds_0) */
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
#ifdef SMOL_TUNING
#pragma unroll
    for (int k = 0; k < KMAX; k++)
        rawpx [k] = lane * 0x01030507u + (uint32_t) k;             /* (ablation without loads: defined values) */
#endif
    /* LIND_V: the lane's two intermediate pixels of the row in flight */
    uint32_t tmp_in [CPL][NCH];
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int i = 0; i < NCH; i++)
            tmp_in [c][i] = 0;
    /* LIND_H: row being written; LIND_V: row being loaded (the fused kernel has no intermediate image) */
    const gbyte *trow = (PASS == LIND_FUSED || hbox) ? nullptr : gptr (J.tmp) + (size_t) (row_first - J.tmp_row0) * J.w_out * TPX;
    /* hbox: the vertical box sums of this tile's rows, four 32-bit channels per column */
    uint32_t hacc [CPL][4];
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int i = 0; i < 4; i++)
            hacc [c][i] = 0;
    auto hbox_flush = [&] ()
    {
#pragma unroll
        for (int c = 0; c < CPL; c++)
            if (ok [c])
                *reinterpret_cast<SMOL_AS_GLOBAL u32x4 *> (gptr (J.tmp) + ((size_t) hb_row * J.w_out + xc [c]) * 16u)
                    = u32x4 { hacc [c][0], hacc [c][1], hacc [c][2], hacc [c][3] };
    };
    if (hbox && hb_empty)
    {
        hbox_flush ();                                  /* (more splits than rows: this one adds nothing) */
        return;
    }

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
        if (dbg & 2u)
            return;
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if ((uint32_t) k * 64u < npx)                           /* wave-uniform */
            {
#ifndef SMOL_KERNEL_EMU
                /* keep the offset a 32-bit value INSIDE the loop: hoisted out of it, its zero-extension becomes a 64-bit register
                 * pair and every load a 64-bit vector add + global_load ... off; as it stands the load is scalar row base +
                 * 32-bit lane offset with no vector instruction at all */
                asm volatile ("" : "+v" (px_off [k]));
#endif
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
        if (dbg & 32u)
            return;
        if (dbg & 8u)
        {
            conv ([] (uint32_t w, uint32_t (&ch) [4]) { ch [0] = (w & 0xffu) << 3; ch [1] = ((w >> 8) & 0xffu) << 3; ch [2] = ((w >> 16) & 0xffu) << 3; ch [3] = w >> 24; });
            return;
        }
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
            if (dbg & 1u)
            {
#pragma unroll
                for (int i = 0; i < NCH; i++)
                    h [c][i] = (tap_ofs [c][0] + r + (uint32_t) i) & 0x07ff07ffu;
                continue;
            }
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

        if (hbox)
        {
            /* this source row into the vertical box sums (wave-uniform weight) */
            const uint32_t wr = (r == hb_y0) ? hb_w1 : ((r == hb_y1) ? hb_lastw : 0x10000u);
#pragma unroll
            for (int c = 0; c < CPL; c++)
            {
                uint32_t ch [4];
                if (W32)
                {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        ch [i] = h [c][i < NCH ? i : 0];
                }
                else
                {
                    ch [0] = h [c][0] & 0xffffu; ch [1] = h [c][0] >> 16;
                    ch [2] = h [c][NCH - 1] & 0xffffu; ch [3] = h [c][NCH - 1] >> 16;
                }
                if (wr == 0x10000u)
                {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        hacc [c][i] += ch [i];
                }
                else
                {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        hacc [c][i] += ((ch [i] * wr) >> 8) & 0x00ffffffu;
                }
            }
            continue;
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
                        acc [c][i] += t;                /* (zeroed behind every output row: a select per sample otherwise) */
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
                        const uint32_t w = (dbg & 16u) ? ((v [c][0] >> 3) & 0xffu) | ((v [c][1] << 5) & 0xff00u) | ((v [c][2] << 13) & 0xff0000u) | (v [c][3] << 24)
                                                       : pack (v [c]);
                        if (ok [c] && !(dbg & 4u))
                        {
                            if (J.bpp_out == 4 && J.out_dword_ok)
                            {
                                if (nt_out)
                                    store_nt32_at (out_row, xc [c] * 4u, w);
                                else
                                {
                                    uint32_t so = xc [c] * 4u;
#ifndef SMOL_KERNEL_EMU
                                    asm volatile ("" : "+v" (so));      /* (scalar row base + 32-bit lane offset: load_row) */
#endif
                                    *reinterpret_cast<SMOL_AS_GLOBAL uint32_t *> (out_row + so) = w;
                                }
                            }
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
#pragma unroll
                for (int c = 0; c < CPL; c++)
#pragma unroll
                    for (int i = 0; i < NCH; i++)
                        acc [c][i] = 0;
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
    if (hbox)
        hbox_flush ();
}

/* ------------------------------------------------------------------------------------------------------------ *
 * k_lind for sources WITHOUT an alpha channel (24-bit rows, gamma-correct): three channels, hand-placed staging  *
 * ------------------------------------------------------------------------------------------------------------ *
 * Round 4's counters on the gamma-correct photograph row (4000 x 3000 RGB8 -> 1600 x 1200, profiles/
 * r04final_sq_srgbphoto.txt): 39 vector lane-operations per SOURCE pixel for an image that has no alpha at all.
 * The disassembly says where they go: staging a pixel is 15 of them -- a 64-bit address add for its load, a byte
 * shift, then per channel a v_bfe + a v_lshl_add for the table address and, after the three look-ups, two packing
 * instructions -- and a tap is twelve packed 16-bit instructions for four channels of which one is the constant
 * 255. Here (same arithmetic as lind_body, smolscale-generic.c:588-614 unpack, :1290-1425 / :1648-2007 filters,
 * :922-935 / :1010-1023 pack):
 *   unpack   the pixel's three bytes in MEMORY order (whether that is R,G,B or B,G,R is folded into the packer's
 *            byte selectors: nothing between unpack and pack tells the channels apart) through the fused table:
 *            per channel ONE SDWA shift (byte k, times two = the table address) and one look-up with the table's
 *            address in its offset field, all look-ups of a row in flight together: 3 + 3 + 1 instructions and a
 *            ds_write_b64 per pixel; the byte shift only in the one tile per row that holds the row's last pixel;
 *   taps     on the packed pair as it comes from LDS, without unpacking: v_mad_u32_u16 picks the half it wants
 *            (op_sel) -- (p * F + q * (256 - F)) >> 8 is two of them and a shift per channel, six 4-cycle and three
 *            2-cycle instructions per pixel-tap against twelve 4-cycle ones; results are one register per channel,
 *            which is how the vertical filter (24-bit multiplies with wave-uniform weights) and the packer want them;
 *   alpha    is the constant 255: neither staged nor filtered; the packer is WP_LIN_O (or the `123' packer's WP_LIN_Q).
 * Tile geometry, schedules and row order are lind_body's. */

/* One look-up in the 16-bit (8-bit) table at offset OFS of the workgroup's table copy, which starts at LDS address 0 (the
 * kernels of this family have no static LDS, and lind_entry traps if that ever changes): the LDS pointer is made from
 * the INTEGER a + OFS, so OFS lands in the instruction's offset field. Through the `lds' symbol the compiler adds the
 * base -- a literal 0 it only learns at link time -- to every address in a vector instruction of its own. a = 2 * index
 * (16-bit entries), a = index (bytes). Plain loads: the compiler keeps every look-up of a row in flight and counts its
 * waits itself. (Round 5 also tried them as asm statements -- ds_read_u16_d16 / _d16_hi would drop two results into
 * the halves of one register and save the packing instruction: on a chip with SRAM ECC a d16 load CLEARS the other
 * half; and an in-flight destination register of an asm load may be copied by the compiler before the wait: wrong
 * bytes on the GPU, right ones in the emulation.) */
#ifndef SMOL_KERNEL_EMU
typedef __attribute__ ((address_space (3))) const uint16_t smol_lds_u16;
typedef __attribute__ ((address_space (3))) const uint8_t smol_lds_u8;
#endif

template <uint32_t OFS>
__device__ __forceinline__ uint32_t smol_tab16 (uint32_t a, const uint8_t *lut_lds)
{
#ifdef SMOL_KERNEL_EMU
    return reinterpret_cast<const uint16_t *> (lut_lds + OFS) [a >> 1];
#else
    (void) lut_lds;
    return *reinterpret_cast<smol_lds_u16 *> (a + OFS);
#endif
}

template <uint32_t OFS>
__device__ __forceinline__ uint32_t smol_tab8 (uint32_t a, const uint8_t *lut_lds)
{
#ifdef SMOL_KERNEL_EMU
    return lut_lds [OFS + a];
#else
    (void) lut_lds;
    return *reinterpret_cast<smol_lds_u8 *> (a + OFS);
#endif
}

/* one tap on a 16-bit half (HI: the upper one) of packed pixels p and q, weights W = F | (256 - F) << 16:
 * p * F + q * (256 - F), not yet shifted (below 2^19 for 11-bit values) */
template <int HI>
__device__ __forceinline__ uint32_t tap16_half (uint32_t p, uint32_t q, uint32_t W)
{
#ifdef SMOL_KERNEL_EMU
    const uint32_t pv = HI ? p >> 16 : p & 0xffffu, qv = HI ? q >> 16 : q & 0xffffu;
    return pv * (W & 0xffffu) + qv * (W >> 16);
#else
    uint32_t t, r;
    if (HI)
    {
        asm ("v_mad_u32_u16 %0, %1, %2, 0 op_sel:[1,1,0,0]" : "=v" (t) : "v" (q), "v" (W));
        asm ("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v" (r) : "v" (p), "v" (W), "v" (t));
    }
    else
    {
        asm ("v_mad_u32_u16 %0, %1, %2, 0 op_sel:[0,1,0,0]" : "=v" (t) : "v" (q), "v" (W));
        asm ("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "=v" (r) : "v" (p), "v" (W), "v" (t));
    }
    return r;
#endif
}

template <int HX, bool NARROW, bool NT = false, bool HV0 = false>
__device__ __forceinline__ void lind3_body (const JobDev &J, uint8_t *buf, const uint8_t *lut_lds,
                                            const uint32_t lane, const uint32_t tile)
{
    constexpr uint32_t NSX = 1u << HX;
    constexpr int CPL = NARROW ? 1 : 2;
    constexpr int KMAX = NARROW ? LIND_KMAX_NARROW : LIND_KMAX;
    constexpr uint32_t OQ = 6656u;                                  /* LutLds::oq, the fused table for opaque pixels */
    const uint32_t W = J.group_rows;

    const uint32_t tiles_x = (J.w_out + W - 1) / W;
    const uint32_t tx = tile % tiles_x;
    const uint32_t ty = tile / tiles_x;
    const uint32_t yrel0 = ty * J.tile_rows;
    if (yrel0 >= J.n_rows)
        return;
    const uint32_t yrel1 = (yrel0 + J.tile_rows < J.n_rows) ? yrel0 + J.tile_rows : J.n_rows;
    const uint32_t x0 = tx * W;
    const uint32_t xn = (x0 + W < J.w_out) ? x0 + W : J.w_out;

    /* ---- horizontal samples of this lane's columns ------------------------------------------------------------ */
    uint32_t xc [CPL];
    bool ok [CPL];
    uint32_t tap_ofs [CPL][NSX], tap_w [CPL][NSX];
#pragma unroll
    for (int c = 0; c < CPL; c++)
    {
        xc [c] = x0 + 64u * c + lane;
        ok [c] = xc [c] < xn;
        const uint32_t x = ok [c] ? xc [c] : xn - 1;
        uint64_t pos = J.frac0_x + (uint64_t) (x << HX) * J.step_x;
#pragma unroll
        for (uint32_t s = 0; s < NSX; s++)
        {
            uint32_t f;
            bilin_tap (pos, J.w_in, tap_ofs [c][s], f);
            if (J.filter_h == F_ONE)
            {
                tap_ofs [c][s] = 0;                                 /* (an axis of ONE pixel: lind_body) */
                f = 256;
            }
            tap_w [c][s] = f | ((256u - f) << 16);
            pos += J.step_x;
        }
    }
    const uint32_t sx0 = smol_readlane (tap_ofs [0][0], 0);
    uint32_t sx1, f_unused;
    bilin_tap (J.frac0_x + (uint64_t) ((xn << HX) - 1) * J.step_x, J.w_in, sx1, f_unused);
    sx1 += 1;
    if (J.filter_h == F_ONE)
        sx1 = 0;
    const uint32_t npx = sx1 - sx0 + 1;
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (uint32_t s = 0; s < NSX; s++)
            tap_ofs [c][s] = (tap_ofs [c][s] - sx0) * 8u;

    /* ---- vertical schedule (lind_body's) ------------------------------------------------------------------------ */
    /* HV0 (and NT, a magnification on both axes): no vertical halving, ONE sample per output row, known at compile time -- no
     * shifts by zero, no sums of one term (gamma-correct x2 331 -> 310 us, 24-bit rows x4 158 -> 137: profiles/r05_lind_hv0.txt).
     * The launcher picks the flavour by the job's halv_v. */
    const uint32_t hv = (NT || HV0) ? 0u : J.halv_v;
    const uint32_t nsy = 1u << hv;
    const uint32_t jy0 = (J.first_row + yrel0) << hv;
    const uint32_t n_samples = (yrel1 - yrel0) << hv;
    uint32_t sched [2] = { 0, 0 };
    const bool v_flat = J.filter_v == F_COPY || J.filter_v == F_ONE;
#pragma unroll
    for (uint32_t q = 0; q < 2; q++)
    {
        uint32_t k = q * 64u + lane;
        if (k >= n_samples)
            k = n_samples - 1;
        uint32_t o, f;
        bilin_tap (J.frac0_y + (uint64_t) (jy0 + k) * J.step_y, J.h_in, o, f);
        if (v_flat)
        {
            o = J.filter_v == F_ONE ? 0u : jy0 + k;
            f = 0;
        }
        sched [q] = o | (f << 16);
    }
    uint32_t e = smol_readlane (sched [0], 0);
    const uint32_t e_last = smol_readlane (sched [(n_samples - 1) >> 6], (n_samples - 1) & 63u);
    const uint32_t vdue = v_flat ? 0u : 1u;
    uint32_t oy = e & 0xffffu, Fy = e >> 16;
    const uint32_t row_first = oy;
    const uint32_t row_last = (e_last & 0xffffu) + vdue;

    /* ---- staging ----------------------------------------------------------------------------------------------- */
    const gbyte *srow = gptr (J.src) + (size_t) row_first * J.src_stride;
    /* the row's last pixel is read one byte early and shifted (its dword would end past the row): only the tile that
     * holds it pays the shift */
    const bool has_last = sx0 + npx == J.w_in;
    uint32_t px_off [KMAX];
    uint32_t px_sh = 0;                                             /* bytes to shift by: 1 in the slots of the last pixel */
#pragma unroll
    for (int k = 0; k < KMAX; k++)
    {
        const uint32_t i = (uint32_t) k * 64u + lane;
        const uint32_t x = sx0 + (i < npx ? i : npx - 1);
        const uint32_t sh = x == J.w_in - 1u ? 1u : 0u;
        px_off [k] = x * 3u - sh;
        px_sh |= sh << k;
    }
    uint32_t rawpx [KMAX];
    const uint32_t one = 1u;
    const uint32_t n_rounds = (npx + 63u) >> 6;                     /* wave-uniform, <= KMAX (host) */

    auto load_row = [&] ()
    {
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if ((uint32_t) k < n_rounds)
            {
#ifdef SMOL_KERNEL_EMU
                __builtin_memcpy (&rawpx [k], srow + px_off [k], 4);
#else
                /* (the offset is made opaque here so that the compiler keeps it a 32-bit value beside the row's scalar base
                 * -- global_load_dword v, v_off, s[base] -- instead of widening it once outside the loop and adding 64-bit
                 * pairs in a vector instruction per load) */
                uint32_t off = px_off [k];
                asm volatile ("" : "+v" (off));
                rawpx [k] = *reinterpret_cast<const SMOL_AS_GLOBAL u32a1 *> (srow + off);
#endif
            }
        srow += J.src_stride;
    };

    auto convert_row = [&] ()
    {
        if (has_last)
        {
            /* (a real branch, taken by one tile per row: the empty asm statement keeps the compiler from turning it into
             * a shift and a select per pixel for everybody) */
            asm volatile ("" ::: "memory");
#pragma unroll
            for (int k = 0; k < KMAX; k++)
                if ((uint32_t) k < n_rounds)
                    rawpx [k] = __builtin_amdgcn_alignbyte (0u, rawpx [k], (px_sh >> k) & 1u);
        }
        /* all look-ups of the row in flight together, then the stores (two passes over the rounds, each behind the
         * wave-uniform "does this round exist") */
        uint32_t t [KMAX][3];
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if ((uint32_t) k < n_rounds)
            {
                t [k][0] = smol_tab16<OQ> (smol_byte_x2<0> (rawpx [k], one), lut_lds);
                t [k][1] = smol_tab16<OQ> (smol_byte_x2<1> (rawpx [k], one), lut_lds);
                t [k][2] = smol_tab16<OQ> (smol_byte_x2<2> (rawpx [k], one), lut_lds);
            }
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if ((uint32_t) k < n_rounds)
                *reinterpret_cast<u32x2a8 *> (buf + ((uint32_t) k * 64u + lane) * 8u) = u32x2a8 { t [k][0] | (t [k][1] << 16), t [k][2] };
    };

    /* ---- the row loop -------------------------------------------------------------------------------------------- */
    /* (the channels travel in the source's memory order: fill_jobdev () has swapped the R / B labels of B,G,R rows on both sides) */
    const JobDev &Jp = J;
    /* 24-bit output: the four pixels of a quad of lanes are twelve bytes = three dwords. Lane j of the quad (j < 3) takes
     * pixel j + 1 from its neighbour (one DPP move), cuts dword j out of the two (one v_perm, selector by j) and stores it
     * at its pixel's address + j (global stores need no alignment on gfx950): one 4-byte store on 48 lanes where there
     * were a 2-byte and a 1-byte store on 64, three instructions of shifting and a 64-bit address add. Quads that reach
     * past the tile's last column keep the byte stores. */
    const uint32_t qj = lane & 3u;
    const uint32_t q_sel = qj == 0 ? 0x04020100u : qj == 1 ? 0x05040201u : 0x06050402u;
    bool q_full [CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++)
        q_full [c] = xc [c] - qj + 3u < xn;                         /* the quad's last pixel is inside the tile */
    /* NT: streaming stores, for magnifications on both axes whose output rows start on 64-byte boundaries (dispatch::lind_nt):
     * gamma-correct x2 358 -> 334 us, x1.1 151 -> 138, RGB8 x4 167 -> 158. A kernel of its own: minifications gain nothing
     * (within 1 %), rows that start anywhere lose (x2.6 on an odd width 99 -> 104: the partial lines at both ends of every
     * row), and as a run-time choice in the common kernel the branch alone cost the arithmetic-bound rows 1-5 %
     * (profiles/r05_lind_nt.txt). */
    constexpr bool nt_out = NT;
    const LutLds lut (lut_lds, LUT_AT_LDS_0);
    const WidePack<WP_LIN_O> wp (Jp, lut);                          /* (its byte selectors; WP_LIN_Q's are the same) */
    const bool quirk = wide_pack_kind (J) == WP_LIN_Q;
    constexpr uint32_t TS = 4608u;                                  /* LutLds::ts, the to_srgb table */
    gbyte *out_row = gptr (J.dst) + (size_t) yrel0 * J.dst_stride;
    uint32_t ks = 0, s_y = 0;
    uint32_t hp [CPL][3], acc [CPL][3];
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int i = 0; i < 3; i++)
            hp [c][i] = acc [c][i] = 0;

    load_row ();

    for (uint32_t r = row_first; r <= row_last; r++)
    {
        uint32_t h [CPL][3];
        smol_wave_sync ();
        convert_row ();
        smol_wave_sync ();
        if (r < row_last)
            load_row ();

#pragma unroll
        for (int c = 0; c < CPL; c++)
        {
            uint32_t hs [3];
#pragma unroll
            for (uint32_t s = 0; s < NSX; s++)
            {
                const u32x4a8 pq = *reinterpret_cast<const u32x4a8 *> (buf + tap_ofs [c][s]);
                const uint32_t t0 = tap16_half<0> (pq.x, pq.z, tap_w [c][s]) >> 8;
                const uint32_t t1 = tap16_half<1> (pq.x, pq.z, tap_w [c][s]) >> 8;
                const uint32_t t2 = tap16_half<0> (pq.y, pq.w, tap_w [c][s]) >> 8;
                hs [0] = s ? hs [0] + t0 : t0;
                hs [1] = s ? hs [1] + t1 : t1;
                hs [2] = s ? hs [2] + t2 : t2;
            }
#pragma unroll
            for (int i = 0; i < 3; i++)
                h [c][i] = hs [i] >> HX;
        }

        while (ks < n_samples && oy + vdue == r)
        {
            const uint32_t Gy = 256u - Fy;
#pragma unroll
            for (int c = 0; c < CPL; c++)
#pragma unroll
                for (int i = 0; i < 3; i++)
                    acc [c][i] += tap32 (hp [c][i], h [c][i], Fy, Gy);      /* (zeroed behind every output row) */
            s_y++;
            ks++;
            if (s_y == nsy)
            {
                /* pack: alpha is 255 -- the inverse for 255 is 1028 = 1024 + 4, (s * 1028) >> 10 = s + (s >> 8), one look-up
                 * per channel (WP_LIN_O); the `123' packer feeds the lane as it is (WP_LIN_Q, smolscale-generic.c:922-935) */
                uint32_t tb [CPL][3];
#pragma unroll
                for (int c = 0; c < CPL; c++)
#pragma unroll
                    for (int i = 0; i < 3; i++)
                    {
                        const uint32_t v = acc [c][i] >> hv;
                        tb [c][i] = smol_tab8<TS> ((quirk ? v : v + (v >> 8)) & 0x7ffu, lut_lds);
                        acc [c][i] = 0;
                    }
#pragma unroll
                for (int c = 0; c < CPL; c++)
                {
                    const uint32_t w = wp.assemble (tb [c][0], tb [c][1], tb [c][2], 0xffu);
                    if (J.bpp_out == 3)
                    {
                        const uint32_t d = __builtin_amdgcn_perm (smol_quad_next (w), w, q_sel);
                        if (q_full [c])
                        {
                            if (qj != 3u)
                            {
#ifdef SMOL_KERNEL_EMU
                                __builtin_memcpy (out_row + (size_t) xc [c] * 3 + qj, &d, 4);
#else
                                if (nt_out)
                                    store_nt32_at (out_row, xc [c] * 3u + qj, d);
                                else
                                    *reinterpret_cast<SMOL_AS_GLOBAL u32a1 *> (out_row + (xc [c] * 3u + qj)) = d;
#endif
                            }
                        }
                        else if (ok [c])
                            wide_store_word (J, out_row, xc [c], w);
                    }
                    else if (ok [c])
                    {
                        if (J.out_dword_ok)
                        {
                            if (nt_out)
                                store_nt32_at (out_row, xc [c] * 4u, w);
                            else
                                *reinterpret_cast<SMOL_AS_GLOBAL uint32_t *> (out_row + (size_t) xc [c] * 4) = w;
                        }
                        else
                            wide_store_word (J, out_row, xc [c], w);
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
#pragma unroll
            for (int i = 0; i < 3; i++)
                hp [c][i] = h [c][i];
    }
}

/* (A flavour of this body WITHOUT gamma for 24-bit rows of the 64bpp class -- a pixel expanded once, two permutes and a
 * ds_write_b64, taps as one 16-byte LDS read and six packed instructions, no tables; "a 24-bit tap reader that expands once
 * per lane instead of per tap" -- was built and measured in round 5: bit-exact, and equal to k_tile at best (4000 x 3000 RGB8
 * -> 1600 x 1200: 118.7 against 117.8-120 us with tiles of two rows, 122-141 with taller ones; slower at every other ratio).
 * profiles/r05_flat3_ab.txt, git 0cc225c.) */

/* (The fused 2-D table L (c, a) for premultiplied sources -- 66 KB of LDS, 16-wave persistent workgroups -- was built and
 * measured in round 5: bit-exact and 1.5 x slower than the chain above, 0.39 -> 0.26 of the roofline; one workgroup per CU
 * costs more than the table saves. profiles/r05_lind_tri_ab.txt, git 3f9fcfe.) */

/* LDS of one wave: the row buffer */
__device__ __forceinline__ uint32_t lind_wave_lds (bool w32, bool narrow = false)
{
    return 64u * (narrow ? LIND_KMAX_NARROW : LIND_KMAX) * (w32 ? 16u : 8u);
}

/* 4 waves per workgroup share one LDS copy of the tables, each wave works on its own tile in its own LDS region */
/* NARROW: the one-column flavour, a kernel of its own (k_lind_batch<HX, IN, true>; as a second body inside the common
 * kernel it cost the wide tiles 3-4 %, as k_boxh's 64-lane flavour did); the host picks it per job: dispatch::lind_narrow */
template <int HX, bool IN24, bool W32, bool NARROW = false, bool NT = false, bool HV0 = false>
__device__ __forceinline__ void lind_entry (const JobDev &J, uint8_t *lds)
{
#ifdef SMOL_KERNEL_EMU
    static uint8_t lut_lds [LUT_LDS_BYTES];
    lut_fill_lds (lut_lds, 0, 1);                       /* emulated: one wave per block */
    if (IN24 && !W32)
        lind3_body<HX, NARROW, NT, HV0> (J, lds, lut_lds, threadIdx.x, blockIdx.x);
    else
        lind_body<HX, IN24, W32, LIND_FUSED, false, NARROW, NT, HV0> (J, lds, lut_lds, threadIdx.x, blockIdx.x);
#else
    /* (readfirstlane: the compiler cannot know that threadIdx.x >> 6 is the same on every lane of a wave, and
     * without it everything derived from the tile index -- every loop bound, every "is this round needed" test --
     * is computed per lane and branched on with exec masks) */
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane ((int) (threadIdx.x >> 6));
    lut_expect_lds_0 (lds);                             /* (the bodies' look-ups carry the tables' LDS addresses as immediates) */
    lut_fill_lds (lds, threadIdx.x, 256);
    __syncthreads ();
    uint8_t *tab_end = lds + ((LUT_LDS_BYTES + 15u) & ~15u);       /* (the host sizes the workgroup's LDS by the flavour too: down_lds_bytes) */
    if (IN24 && !W32)
    {
        lind3_body<HX, NARROW, NT, HV0> (J, tab_end + wave * lind_wave_lds (false, NARROW), lds, threadIdx.x & 63u, blockIdx.x * 4u + wave);
    }
    else
        lind_body<HX, IN24, W32, LIND_FUSED, false, NARROW, NT, HV0> (J, tab_end + wave * lind_wave_lds (W32, NARROW), lds, threadIdx.x & 63u, blockIdx.x * 4u + wave);
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
    lut_expect_lds_0 (lds);
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
    lut_expect_lds_0 (lds);
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
    if (dispatch::twopass_hbox (J))
        return J.linear ? 2 : 3;        /* (box sums in pass 1's registers: the 19-bit channels are never narrowed) */
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
template <int HX, int IN, bool NARROW = false, bool NT = false, bool HV0 = false>
__global__ void __launch_bounds__ (256)
k_lind_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.y);
    lind_entry<HX, IN == 1, IN == 2, NARROW, NT, HV0> (J, smol_dyn_lds);
}

template <int HX, int IN, bool NARROW = false, bool NT = false, bool HV0 = false>
__global__ void __launch_bounds__ (256)
k_lind_one (const JobDev J)
{
    lind_entry<HX, IN == 1, IN == 2, NARROW, NT, HV0> (J, smol_dyn_lds);
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
