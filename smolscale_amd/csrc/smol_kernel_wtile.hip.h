/* smol_kernel_wtile.hip.h -- k_wtile: the tiled kernel for 128bpp-class jobs
 * (unassociated -> unassociated "premul16", sRGB-linear, i.e. the reference's
 * 128bpp storage: smolscale-generic.c:470-752 unpack, :1324-1418 horizontal,
 * :1734-1996 vertical, :912-1164 pack) with BILINEAR 0-2H / COPY / ONE on both
 * axes. Box axes of this class stay on k_general.
 *
 * Same wave-autonomous structure as k_tile (64-lane workgroup = 64 output columns
 * x tile_rows rows, double-buffered LDS-DMA staging, vertical filter streamed in
 * registers) with one extra phase: the unpack of this class is expensive (inverse
 * table, sRGB table, alpha products), so every staged source pixel is converted
 * ONCE into four 16-bit internal channels in LDS, and the taps read those. Tap
 * arithmetic is per channel in 32-bit registers; with non-negative operands
 *     q + (((p - q) * F) >> 8)  ==  (p * F + q * (256 - F)) >> 8
 * holds for this class too (values stay below 2^17). */

#pragma once

namespace smolhip {

struct Px4 { uint32_t c [4]; };     /* R, G, B, alpha lane */

/* Converted pixels are four 16-bit channels (8 bytes), except for premul16 on
 * linear light, whose colour lanes need 19 bits: there they are four 32-bit
 * channels (16 bytes). `wide32' is wave-uniform. */
__device__ __forceinline__ Px4 px4_load (const uint8_t *conv, uint32_t idx, bool wide32)
{
    Px4 r;
    if (wide32)
    {
        const uint32_t *p = reinterpret_cast<const uint32_t *> (conv + (size_t) idx * 16);
        r.c [0] = p [0]; r.c [1] = p [1]; r.c [2] = p [2]; r.c [3] = p [3];
    }
    else
    {
        const uint32_t *p = reinterpret_cast<const uint32_t *> (conv + (size_t) idx * 8);
        const uint32_t lo = p [0], hi = p [1];
        r.c [0] = lo & 0xffffu; r.c [1] = lo >> 16; r.c [2] = hi & 0xffffu; r.c [3] = hi >> 16;
    }
    return r;
}

__device__ __forceinline__ Px4 px4_tap (const Px4 &p, const Px4 &q, uint32_t F)
{
    const uint32_t G = 256u - F;
    Px4 r;
#pragma unroll
    for (int i = 0; i < 4; i++)
        r.c [i] = (p.c [i] * F + q.c [i] * G) >> 8;
    return r;
}

template <int HM>
__device__ __forceinline__ void wtile_body (const JobDev &J, uint8_t *sbuf)
{
    constexpr bool HPAIR = (HM != 0);
    constexpr uint32_t NSX = HPAIR ? (1u << (HPAIR ? HM - 1 : 0)) : 1u;
    constexpr uint32_t HX = HPAIR ? (uint32_t) (HM - 1) : 0u;

    const uint32_t bpp = J.bpp_in;
    const uint32_t lane = threadIdx.x;
    const uint32_t tiles_x = (J.w_out + TILE_COLS - 1) / TILE_COLS;
    const uint32_t tx = blockIdx.x % tiles_x;
    const uint32_t ty = blockIdx.x / tiles_x;
    const uint32_t yrel0 = ty * J.tile_rows;

    if (yrel0 >= J.n_rows)
        return;

    const uint32_t yrel1 = (yrel0 + J.tile_rows < J.n_rows) ? yrel0 + J.tile_rows : J.n_rows;
    const uint32_t x0 = tx * TILE_COLS;
    const uint32_t xn = (x0 + TILE_COLS < J.w_out) ? x0 + TILE_COLS : J.w_out;
    const uint32_t x = x0 + lane;
    const bool active = x < xn;
    const uint32_t xc = active ? x : xn - 1;

    /* ---- horizontal taps (pixel index inside the span) ------------------------- */
    uint32_t tap_idx [NSX], tap_f [NSX];
    uint32_t sx0, sx1;

    if (HPAIR)
    {
        uint64_t pos = J.frac0_x + (uint64_t) (xc << HX) * J.step_x;
#pragma unroll
        for (uint32_t s = 0; s < NSX; s++)
        {
            bilin_tap (pos, J.w_in, tap_idx [s], tap_f [s]);
            pos += J.step_x;
        }
        uint32_t f_unused;
        bilin_tap (J.frac0_x + (uint64_t) (x0 << HX) * J.step_x, J.w_in, sx0, f_unused);
        bilin_tap (J.frac0_x + (uint64_t) ((xn << HX) - 1) * J.step_x, J.w_in, sx1, f_unused);
        sx1 += 1;
    }
    else
    {
        tap_idx [0] = (J.filter_h == F_ONE) ? 0u : xc;
        tap_f [0] = 256u;
        sx0 = (J.filter_h == F_ONE) ? 0u : x0;
        sx1 = (J.filter_h == F_ONE) ? 0u : xn - 1;
    }
#pragma unroll
    for (uint32_t s = 0; s < NSX; s++)
        tap_idx [s] -= sx0;

    const uint32_t npx = sx1 - sx0 + 1;                 /* pixels of the span (<= J.span_px) */
    const uint32_t g0 = sx0 * bpp;
    const uint32_t g1 = (sx1 + 1) * bpp;
    const uint32_t row_pitch = J.span_chunks * 16u;
    const uint32_t raw_bytes = J.group_rows * row_pitch;
    const bool wide32 = J.p16 && J.linear;
    const uint32_t conv_pitch = J.span_px * (wide32 ? 16u : 8u);
    uint8_t *conv = sbuf + 2u * raw_bytes;
    /* workgroup-private copy of the parity tables, behind the converted rows */
    uint8_t *lut_base = conv + ((J.group_rows * conv_pitch + 15u) & ~15u);
    lut_fill_lds (lut_base, lane, TILE_COLS);
    __syncthreads ();
    const LutLds lut (lut_base);

    /* ---- vertical schedule (see k_tile) ------------------------------------------ */
    const bool vpair = (J.filter_v == F_BILINEAR);
    const uint32_t hv = vpair ? J.halv_v : 0u;
    const uint32_t nsy = 1u << hv;
    const uint32_t vlag = vpair ? 1u : 0u;
    const uint32_t y_first = J.first_row + yrel0;
    const uint32_t jy0 = y_first << hv;
    const uint32_t n_samples = (yrel1 - yrel0) << hv;
    uint32_t sched [2];
#pragma unroll
    for (uint32_t q = 0; q < 2; q++)
    {
        uint32_t k = q * 64u + lane;
        if (k >= n_samples)
            k = n_samples - 1;
        uint32_t o, f;
        tile_vsample (J, jy0 + k, o, f);
        sched [q] = o | (f << 16);
    }
    uint32_t ks = 0, s_y = 0;
    uint32_t e = smol_readlane (sched [0], 0);
    uint32_t oy = e & 0xffffu, Fy = e >> 16;
    const uint32_t row_first = oy;
    const uint32_t e_last = smol_readlane (sched [(n_samples - 1) >> 6], (n_samples - 1) & 63u);
    const uint32_t row_last = (e_last & 0xffffu) + vlag;

    Px4 hprev, acc;
#pragma unroll
    for (int i = 0; i < 4; i++) { hprev.c [i] = 0; acc.c [i] = 0; }

    const uintptr_t src_base = (uintptr_t) J.src;
    uint8_t *out_row = J.dst + (size_t) yrel0 * J.dst_stride;

    uint32_t gfirst = row_first;
    uint32_t nrows = (row_last - gfirst + 1 < J.group_rows) ? row_last - gfirst + 1 : J.group_rows;
    uint32_t cur = 0;

    tile_stage<true, false> (J, sbuf, lane, gfirst, nrows, g0, g1);

    while (true)
    {
        const uint8_t *raw = sbuf + cur * raw_bytes;

        smol_wait_vmcnt0 ();
        __syncthreads ();               /* raw group landed; previous conv rows consumed */
        {
            const uint32_t nfirst = gfirst + nrows;
            if (nfirst <= row_last)
            {
                const uint32_t nn = (row_last - nfirst + 1 < J.group_rows) ? row_last - nfirst + 1 : J.group_rows;
                tile_stage<true, false> (J, sbuf + (cur ^ 1u) * raw_bytes, lane, nfirst, nn, g0, g1);
            }
        }

        /* ---- convert: each staged pixel once -> four 16-bit internal channels ------ */
        for (uint32_t i = 0; i < nrows; i++)
        {
            const uint32_t mis = (uint32_t) ((src_base + (size_t) (gfirst + i) * J.src_stride + g0) & 15);
            const uint8_t *rowb = raw + i * row_pitch;
            uint8_t *crow = conv + i * conv_pitch;

            for (uint32_t p = lane; p < npx; p += TILE_COLS)
            {
                const uint32_t w = lds_load_unaligned (rowb, mis + p * bpp);
                uint32_t c [4];
                unpack_word<true> (J, w, c, lut);
                if (wide32)
                {
                    uint32_t *dstp = reinterpret_cast<uint32_t *> (crow + (size_t) p * 16);
                    dstp [0] = c [0]; dstp [1] = c [1]; dstp [2] = c [2]; dstp [3] = c [3];
                }
                else
                {
                    uint32_t *dstp = reinterpret_cast<uint32_t *> (crow + (size_t) p * 8);
                    dstp [0] = (c [0] & 0xffffu) | (c [1] << 16);
                    dstp [1] = (c [2] & 0xffffu) | (c [3] << 16);
                }
            }
        }
        __syncthreads ();

        /* ---- consume ------------------------------------------------------------------ */
        for (uint32_t i = 0; i < nrows; i++)
        {
            const uint32_t r = gfirst + i;
            const uint8_t *crow = conv + i * conv_pitch;
            Px4 h;

            if (HM == 0)
            {
                h = px4_load (crow, tap_idx [0], wide32);
            }
            else
            {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    h.c [k] = 0;
#pragma unroll
                for (uint32_t s = 0; s < NSX; s++)
                {
                    const Px4 t = px4_tap (px4_load (crow, tap_idx [s], wide32), px4_load (crow, tap_idx [s] + 1, wide32), tap_f [s]);
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        h.c [k] += t.c [k];
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                    h.c [k] >>= HX;
            }

            while (ks < n_samples && oy + vlag == r)
            {
                Px4 t = h;
                if (vpair)
                    t = px4_tap (hprev, h, Fy);
#pragma unroll
                for (int k = 0; k < 4; k++)
                    acc.c [k] = (s_y == 0) ? t.c [k] : acc.c [k] + t.c [k];
                s_y++;
                ks++;
                if (s_y == nsy)
                {
                    uint32_t o [4];
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        o [k] = acc.c [k] >> hv;
                    if (active)
                        store_px<true> (J, out_row, x, o, lut);
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
            hprev = h;
        }

        gfirst += nrows;
        if (gfirst > row_last)
            break;
        nrows = (row_last - gfirst + 1 < J.group_rows) ? row_last - gfirst + 1 : J.group_rows;
        cur ^= 1u;
    }
}

__device__ __forceinline__ void wtile_dispatch (const JobDev &J, uint8_t *sbuf)
{
    switch ((J.tile_variant >> 2) & 7u)
    {
        case 0:  wtile_body<0> (J, sbuf); break;
        case 1:  wtile_body<1> (J, sbuf); break;
        case 2:  wtile_body<2> (J, sbuf); break;
        default: wtile_body<3> (J, sbuf); break;
    }
}

__global__ void __launch_bounds__ (64)
k_wtile_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.y);
    wtile_dispatch (J, smol_dyn_lds);
}

__global__ void __launch_bounds__ (64)
k_wtile_one (const JobDev J)
{
    wtile_dispatch (J, smol_dyn_lds);
}

} /* namespace smolhip */
