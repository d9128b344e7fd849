/* smol_kernel_hspan.hip.h -- k_hspan: row-span kernel for "width-only" minification of the 64bpp
 * class: vertical COPY, horizontal BILINEAR with 0-2 halvings at a ratio that is (nearly) the
 * power of two the halvings imply -- BASELINE config 4 (65535 x 4096 RGB8 -> 8192 x 4096 RGBA8:
 * BILINEAR_2H x COPY, every output pixel = 4 samples of 2 adjacent source pixels, weights drifting
 * through all 256 values along the row). Reference: unpack smolscale-generic.c:353-367 / 369-464,
 * horizontal filter :1290-1322, 1363-1387, 1420-1425, pack :758-906; rows are independent (:2262-2290).
 *
 * Why not k_tile: with a vertical copy there is no vertical state, so 2-D tiles only chop the rows
 * into short segments. Measured floor of the access patterns on config 4 (tools/microbench/spanbw.hip,
 * stagebw.hip, gatherbw.hip; profiles/r02c_*): 64-column x 16-row tiles with a row pipeline 166-196 us,
 * per-lane gathers straight from global memory 200-270 us, ROW SPANS 156-160 us -- against 143 us for
 * a plain copy of the same bytes.
 *
 * Shape. A workgroup = 4 waves side by side on one row; a wave owns 256 adjacent output columns, a
 * lane 4 of them (one 16-byte non-temporal store per lane and row, 4 KiB contiguous per workgroup).
 * The wave walks a band of rows. Per row it
 *   1. has the row's span -- 256 * 2^(h+1) source pixels, contiguous -- in registers, loaded by
 *      coalesced 16-byte ALIGNED non-temporal loads (whatever the row alignment or pixel size), NL per
 *      lane, all in flight together; the NEXT row's loads are issued as soon as these have been copied
 *      to the wave's private LDS image, so they fly during the arithmetic;
 *   2. reads its own window back from LDS. Regular waves (every sample takes two fresh pixels:
 *      ofs_j = ofs_0 + 2j -- all but the one wave per row where the sampling grid slips by a pixel):
 *      whole 16-byte frames (ds_read_b128, 2-way bank conflicts at most), one wave-uniform byte
 *      shift, then every pixel sits at a compile-time position: v_perm_b32 with literal selectors
 *      splits 24-bit pixels into packed 16-bit lanes. Irregular waves read sample by sample
 *      (k_tile's reader);
 *   3. taps in packed 16-bit lanes, (p*F + q*(256-F)) >> 8 -- exact for this class -- with the
 *      per-sample weights kept in registers for the whole band (they depend on the column only:
 *      the low 32 bits of the 32.32 position advance by a constant, no 64-bit arithmetic per row).
 * No barriers: the LDS image is private to the wave and LDS operations of a wave complete in order. */

#pragma once

namespace smolhip {

constexpr uint32_t HSPAN_WAVES = 4;         /* waves per workgroup, side by side on a row      */
constexpr uint32_t HSPAN_CPL = 4;           /* output columns per lane                         */
constexpr uint32_t HSPAN_TW = 64 * HSPAN_CPL;

template <int BPP, int HX>
struct HSpanCfg
{
    static constexpr uint32_t NS = 1u << HX;                    /* samples per output column      */
    static constexpr uint32_t WPX = HSPAN_CPL * 2u * NS;        /* regular window, pixels per lane */
    static constexpr uint32_t WB = WPX * BPP;                   /* ... bytes (multiple of 16)     */
    static constexpr uint32_t NW = WB / 4u;                     /* ... dwords                     */
    static constexpr uint32_t NF = WB / 16u + 1u;               /* 16-byte frames a window can touch */
    static constexpr uint32_t NL = (64u * WB + 1023u) / 1024u + 1u;   /* 16-byte loads per lane and row */
    static constexpr uint32_t LDS_PER_WAVE = NL * 1024u + 32u;
    static_assert (WB % 16u == 0, "a lane's window must start on a frame boundary for every lane");
};

/* (p*F + q*G) >> 8 per 16-bit lane, G2 = {G, G}, F = 256 - G */
__device__ __forceinline__ Px2 hspan_tap (Px2 p, Px2 q, uint32_t G2)
{
    /* 0x0100 - G >= 1 in each lane: one 32-bit subtraction serves both lanes */
    const u16x2 g = as_u16x2 (G2), f = as_u16x2 (0x01000100u - G2);
    Px2 r;
    r.lo = as_u32 ((as_u16x2 (p.lo) * f + as_u16x2 (q.lo) * g) >> 8);
    r.hi = as_u32 ((as_u16x2 (p.hi) * f + as_u16x2 (q.hi) * g) >> 8);
    return r;
}

/* Sample S (of this lane's window) out of wc[], the window dwords from sample pair BASE on:
 * pixels 2S and 2S+1, pixel i at bytes BPP*i of the window */
template <int BPP, int S>
__device__ __forceinline__ void hspan_window_pair (const JobDev &J, const uint32_t *wc, Px2 &p, Px2 &q)
{
    if (BPP == 4)
    {
        const uint32_t wp = wc [2 * S], wq = wc [2 * S + 1];
        if (J.unassoc_in)
        {
            p = premul_px2 (wp, J.pos_in [3]);
            q = premul_px2 (wq, J.pos_in [3]);
        }
        else
        {
            p = px2_from_word (wp);
            q = px2_from_word (wq);
        }
    }
    else
    {
        /* two samples = 12 bytes = 3 dwords; selector 0x0c -> 0x00, 0x0d -> 0xff (the alpha lane) */
        constexpr int M = S >> 1;
        if ((S & 1) == 0)
        {
            const uint32_t a = wc [3 * M], b = wc [3 * M + 1];          /* p = a.0-2, q = a.3 b.0-1 */
            p.lo = __builtin_amdgcn_perm (b, a, 0x0c020c00u);
            p.hi = __builtin_amdgcn_perm (b, a, 0x0c0d0c01u);
            q.lo = __builtin_amdgcn_perm (b, a, 0x0c050c03u);
            q.hi = __builtin_amdgcn_perm (b, a, 0x0c0d0c04u);
        }
        else
        {
            const uint32_t a = wc [3 * M + 1], b = wc [3 * M + 2];      /* p = a.2-3 b.0, q = b.1-3 */
            p.lo = __builtin_amdgcn_perm (b, a, 0x0c040c02u);
            p.hi = __builtin_amdgcn_perm (b, a, 0x0c0d0c03u);
            q.lo = __builtin_amdgcn_perm (b, a, 0x0c070c05u);
            q.hi = __builtin_amdgcn_perm (b, a, 0x0c0d0c06u);
        }
    }
}

/* weight byte S & 3 of gw -> {G, G} in 16-bit lanes */
template <int S>
__device__ __forceinline__ uint32_t hspan_g2 (uint32_t gw)
{
    constexpr uint32_t k = (uint32_t) (S & 3);
    return __builtin_amdgcn_perm (gw, gw, 0x0c000c00u | k | (k << 16));
}

/* One row of a REGULAR wave: this lane's window starts `sh' + 4 * K bytes into the frames at
 * `fr' (K compile-time: the caller switches on it once per row, wave-uniformly). Column by
 * column, so that only one column's pixels are unpacked at a time. */
/* VAR: experiment bits (tools/microbench/hspan_ab.hip): 1 = no prefetch of the next row, 2 = read all
 * frames up front, 4 = no scheduling barrier between columns. The library instantiates HSPAN_VAR. */
constexpr int HSPAN_VAR = 0;

template <int BPP, int HX, int K, int VAR>
__device__ __forceinline__ void hspan_regular_row (const JobDev &J, const u32x4 *fr, uint32_t sh,
                                                   const uint32_t *gw, Px2 (&o) [HSPAN_CPL])
{
    using C = HSpanCfg<BPP, HX>;
    constexpr int NS = 1 << HX;
    constexpr int NWC = (int) C::NW / (int) HSPAN_CPL;          /* window dwords per column */
    uint32_t d [C::NF * 4u];
    int have = 0;                                               /* frames read so far (compile-time after unrolling) */
#pragma unroll
    for (int c = 0; c < (int) HSPAN_CPL; c++)
    {
        /* this column needs window dwords c*NWC .. (c+1)*NWC - 1, i.e. frame dwords up to (c+1)*NWC + K */
        const int need = (VAR & 2) ? (int) C::NF : (c * NWC + NWC + K) / 4 + 1;
#pragma unroll
        for (int i = 0; i < (int) C::NF; i++)
            if (i >= have && i < need)
            {
                const u32x4 t = fr [i];
                d [4 * i] = t.x; d [4 * i + 1] = t.y; d [4 * i + 2] = t.z; d [4 * i + 3] = t.w;
            }
        have = need;
        uint32_t wc [NWC];
#pragma unroll
        for (int n = 0; n < NWC; n++)
            wc [n] = __builtin_amdgcn_alignbyte (d [c * NWC + n + K + 1], d [c * NWC + n + K], sh);
        Px2 acc = { 0, 0 };
        if (VAR & 8)
        {
            /* timing ablation: no unpacking, no taps */
#pragma unroll
            for (int n = 0; n < NWC; n++) { acc.lo ^= wc [n]; acc.hi += wc [n] >> 3; }
            o [c] = acc;
            continue;
        }
        /* NS samples: written out for the three possible counts so that every index is a constant */
        {
            Px2 p, q;
            hspan_window_pair<BPP, 0> (J, wc, p, q);
            acc = hspan_tap (p, q, hspan_g2<0> (gw [(c * NS + 0) >> 2] >> (HX == 0 ? 8 * c : (HX == 1 ? 16 * (c & 1) : 0))));
        }
        if (NS >= 2)
        {
            Px2 p, q;
            hspan_window_pair<BPP, (NS >= 2) ? 1 : 0> (J, wc, p, q);
            acc = px2_add (acc, hspan_tap (p, q, hspan_g2<1> (gw [(c * NS + 1) >> 2] >> (HX == 1 ? 16 * (c & 1) : 0))));
        }
        if (NS >= 4)
        {
            Px2 p, q;
            hspan_window_pair<BPP, (NS >= 4) ? 2 : 0> (J, wc, p, q);
            acc = px2_add (acc, hspan_tap (p, q, hspan_g2<2> (gw [(NS >= 4) ? c : 0])));
            hspan_window_pair<BPP, (NS >= 4) ? 3 : 0> (J, wc, p, q);
            acc = px2_add (acc, hspan_tap (p, q, hspan_g2<3> (gw [(NS >= 4) ? c : 0])));
        }
        o [c] = acc;
#ifndef SMOL_KERNEL_EMU
        if (!(VAR & 4))
            __builtin_amdgcn_sched_barrier (0);     /* keep the columns apart: register pressure */
#endif
    }
}

template <int BPP, int HX, int OUT, int VAR = HSPAN_VAR>
__device__ __forceinline__ void hspan_body (const JobDev &J, uint8_t *lds_all)
{
    using C = HSpanCfg<BPP, HX>;
    constexpr int TAP = (BPP == 3) ? TAP_U3 : TAP_U4;
    constexpr uint32_t NS = C::NS;
    constexpr uint32_t NSL = HSPAN_CPL * NS;                    /* samples per lane */

    const uint32_t lane = threadIdx.x & 63u;
#ifdef SMOL_KERNEL_EMU
    const uint32_t wave = threadIdx.x >> 6;
#else
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane ((int) (threadIdx.x >> 6));
#endif
    uint8_t *img = lds_all + wave * C::LDS_PER_WAVE;

    const uint32_t col_blocks = (J.w_out + HSPAN_TW * HSPAN_WAVES - 1u) / (HSPAN_TW * HSPAN_WAVES);
    const uint32_t band = blockIdx.x / col_blocks;
    const uint32_t cb = blockIdx.x - band * col_blocks;
    const uint32_t yrel0 = band * J.tile_rows;
    const uint32_t yrel1 = (yrel0 + J.tile_rows < J.n_rows) ? yrel0 + J.tile_rows : J.n_rows;
    const uint32_t x0 = (cb * HSPAN_WAVES + wave) * HSPAN_TW;
    /* a wave past the right edge still walks the rows (it shares the loop shape with its
     * neighbours) but loads, computes and stores nothing */
    const bool wave_live = (yrel0 < J.n_rows) && (x0 < J.w_out);
    const uint32_t xn = wave_live ? ((x0 + HSPAN_TW < J.w_out) ? x0 + HSPAN_TW : J.w_out) : x0 + 1u;
    const uint32_t xl = x0 + lane * HSPAN_CPL;

    /* ---- the wave's span and whether it is regular (wave-uniform) ------------------------------ */
    const uint32_t j_first = x0 << HX, j_last = ((xn << HX) - 1u);
    const uint64_t pos_first = J.frac0_x + (uint64_t) j_first * J.step_x;
    const uint64_t pos_last = J.frac0_x + (uint64_t) j_last * J.step_x;
    uint32_t sx0, sx1, f_unused;
    bilin_tap (pos_first, J.w_in, sx0, f_unused);
    bilin_tap (pos_last, J.w_in, sx1, f_unused);
    /* every step between consecutive samples is 1 or 2 (or 2 or 3) pixels for the ratios this kernel
     * takes, so "the last sample sits 2 * (n - 1) pixels after the first" means every step is 2;
     * a clamped last sample (right image edge, bilin_tap) has to go the careful way too */
    const bool regular = wave_live && (xn == x0 + HSPAN_TW)
                         && (sx1 - sx0 == 2u * (j_last - j_first))
                         && ((uint32_t) (pos_last >> 32) < J.w_in - 1u);
    sx1 += 1u;
    const uint32_t g0 = sx0 * BPP;                              /* first byte of the span in a row */
    const uint32_t span_bytes = (sx1 + 1u) * BPP - g0;

    /* ---- per-sample weights of this lane's columns: constant over the band. G = 256 - F = the 8
     * fraction bits of the position (0..255), four samples per register ------------------------- */
    uint32_t gw [(NSL + 3u) / 4u];
    {
        const uint32_t xc = (xl < xn) ? xl : xn - 1u;
        const uint64_t pos = J.frac0_x + (uint64_t) (xc << HX) * J.step_x;
        uint32_t lo = (uint32_t) pos;
        const uint32_t step_lo = (uint32_t) J.step_x;
#pragma unroll
        for (uint32_t s = 0; s < (NSL + 3u) / 4u; s++)
            gw [s] = 0;
#pragma unroll
        for (uint32_t s = 0; s < NSL; s++)
        {
            gw [s >> 2] |= (lo >> 24) << (8u * (s & 3u));
            lo += step_lo;
        }
    }

    const uintptr_t src_base = (uintptr_t) J.src;
    uintptr_t ra = src_base + (size_t) (J.first_row + yrel0) * J.src_stride + g0;   /* first needed byte */
    uint8_t *out_row = J.dst + (size_t) yrel0 * J.dst_stride;
    const bool interior = (xn == x0 + HSPAN_TW) && J.out_vec16_ok;
    const bool full24 = (xn == x0 + HSPAN_TW) && J.out3_ok;

    u32x4 v [C::NL];
    auto issue = [&] (uintptr_t row_first_byte) {
        const uintptr_t a0 = row_first_byte & ~(uintptr_t) 15;
        const uint32_t nch = (((uint32_t) row_first_byte & 15u) + span_bytes + 15u) >> 4;
#pragma unroll
        for (uint32_t i = 0; i < C::NL; i++)
        {
            const uint32_t k = i * 64u + lane;
            if (k < nch)
                v [i] = __builtin_nontemporal_load (reinterpret_cast<const SMOL_AS_GLOBAL u32x4 *> (a0 + (uintptr_t) k * 16));
        }
    };

    if (wave_live)
        issue (ra);

    for (uint32_t yr = yrel0; yr < yrel1; yr++)
    {
        const uint32_t mis = (uint32_t) ra & 15u;
        if ((VAR & 1) && wave_live && yr > yrel0)
            issue (ra);
        if (wave_live)
        {
            const uint32_t nch = (mis + span_bytes + 15u) >> 4;
#pragma unroll
            for (uint32_t i = 0; i < C::NL; i++)
            {
                const uint32_t k = i * 64u + lane;
                if (k < nch)
                    *reinterpret_cast<u32x4 *> (img + (size_t) k * 16) = v [i];
            }
            /* the registers are free again: the next row's loads fly during the arithmetic below */
            if (!(VAR & 1) && yr + 1u < yrel1)
                issue (ra + J.src_stride);
        }
        smol_wave_sync ();

        Px2 o [HSPAN_CPL];
        if (regular)
        {
            /* frames lane*WB .. of the image, then one wave-uniform shift by `mis' bytes */
            const u32x4 *fr = reinterpret_cast<const u32x4 *> (img + (size_t) lane * C::WB);
            const uint32_t sh = mis & 3u;
            switch (mis >> 2)
            {
                case 0:  hspan_regular_row<BPP, HX, 0, VAR> (J, fr, sh, gw, o); break;
                case 1:  hspan_regular_row<BPP, HX, 1, VAR> (J, fr, sh, gw, o); break;
                case 2:  hspan_regular_row<BPP, HX, 2, VAR> (J, fr, sh, gw, o); break;
                default: hspan_regular_row<BPP, HX, 3, VAR> (J, fr, sh, gw, o); break;
            }
        }
        else if (wave_live)
        {
            /* sample by sample, positions from the closed form (edge clamps included); rare (one
             * wave per row), so rolled loops: small code, few registers */
#pragma unroll 1
            for (uint32_t c = 0; c < HSPAN_CPL; c++)
            {
                const uint32_t xc = (xl + c < xn) ? xl + c : xn - 1u;
                uint64_t pos = J.frac0_x + (uint64_t) (xc << HX) * J.step_x;
                Px2 acc = { 0, 0 };
#pragma unroll 1
                for (uint32_t s = 0; s < NS; s++)
                {
                    uint32_t ofs, F;
                    bilin_tap (pos, J.w_in, ofs, F);
                    pos += J.step_x;
                    Px2 p, q;
                    tile_read_pair<TAP, true> (J, img, mis + (ofs - sx0) * BPP, p, q);
                    acc = px2_add (acc, tap2 (p, q, F));
                }
                /* o [c] with a run-time c would live in scratch memory: select */
                if (c == 0) o [0] = acc;
                if (c == 1) o [1] = acc;
                if (c == 2) o [2] = acc;
                if (c == 3) o [3] = acc;
            }
        }
        smol_wave_sync ();

        if (wave_live)
        {
            if (HX)
            {
#pragma unroll
                for (uint32_t c = 0; c < HSPAN_CPL; c++)
                    o [c] = px2_shr (o [c], HX);
            }
            tile_store_cols<OUT, (int) HSPAN_CPL> (J, out_row, xl, xn, interior, full24, o);
        }
        ra += J.src_stride;
        out_row += J.dst_stride;
    }
}

template <int BPP, int HX, int VAR = HSPAN_VAR>
__device__ __forceinline__ void hspan_dispatch_out (const JobDev &J, uint8_t *lds)
{
    switch (J.tile_variant & 3u)
    {
        case OUT_P4: hspan_body<BPP, HX, OUT_P4, VAR> (J, lds); break;
        case OUT_U4: hspan_body<BPP, HX, OUT_U4, VAR> (J, lds); break;
        default:     hspan_body<BPP, HX, OUT_ANY, VAR> (J, lds); break;
    }
}

#ifdef SMOL_KERNEL_EMU
__global__ void k_hspan_one (const JobDev J)
{
    if (J.bpp_in == 3)
    {
        if (J.halv_h == 1) hspan_dispatch_out<3, 1> (J, smol_dyn_lds);
        else               hspan_dispatch_out<3, 2> (J, smol_dyn_lds);
    }
    else
    {
        if (J.halv_h == 0)      hspan_dispatch_out<4, 0> (J, smol_dyn_lds);
        else if (J.halv_h == 1) hspan_dispatch_out<4, 1> (J, smol_dyn_lds);
        else                    hspan_dispatch_out<4, 2> (J, smol_dyn_lds);
    }
}
#else
template <int BPP, int HX>
__global__ void __launch_bounds__ (64 * HSPAN_WAVES, (BPP == 4 && HX == 2) ? 3 : 4)
k_hspan_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.y);
    hspan_dispatch_out<BPP, HX> (J, smol_dyn_lds);
}

template <int BPP, int HX>
__global__ void __launch_bounds__ (64 * HSPAN_WAVES, (BPP == 4 && HX == 2) ? 3 : 4)
k_hspan_one (const JobDev J)
{
    hspan_dispatch_out<BPP, HX> (J, smol_dyn_lds);
}
#endif

} /* namespace smolhip */
