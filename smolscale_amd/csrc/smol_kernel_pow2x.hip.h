/* smol_kernel_pow2x.hip.h -- k_pow2x: exact 2:1 / 4:1 / 8:1 (same ratio on both
 * axes) for EVERY pixel-type pair and both storage classes. Included at the end
 * of smol_kernels.hip.h.
 *
 * k_pow2 (smol_kernels.hip.h) streams 32-bit premultiplied pixels only. This is
 * the same streaming shape -- 16-byte loads contiguous per lane, all loads of a
 * lane's source block issued before any arithmetic, no staging in LDS, every
 * source byte read once -- with the conversions of the other pixel types done in
 * registers:
 *
 *   IN = PX_N32  32-bit input, 64bpp-class math. Unassociated input is
 *                premultiplied per pixel word (smolscale-generic.c:238-244,
 *                415-464); all four 8-bit channels stay packed in one register
 *                and every tap is a per-byte floor average, as in k_pow2.
 *   IN = PX_N24  24-bit input, 64bpp class: the lane's unit is a 12-byte load =
 *                4 pixels (still contiguous across the wave: 768 bytes per load
 *                instruction), re-cut into 4 pixel words (alpha = 0xff, :353-413)
 *                with v_alignbyte.
 *   IN = PX_W32 / PX_W24 / PX_W16
 *                128bpp class (premul16 and/or sRGB-linear, :470-752): pixels are
 *                unpacked to four 32-bit channels by WideUnpack / WidePack with the
 *                flavour chosen once per lane (smol_kernels.hip.h), tables read from
 *                an LDS copy.
 *
 * With w_in == w_out << (H+1) every sample sits at an even offset with F = 128
 * (closed form of :24-52), so for any channel width a tap
 *      q + (((p - q) * 128) >> 8)  ==  floor ((p + q) / 2)
 * and a halving is (sum of 2^H taps) >> H (:1363-1387, :1684-1811); lanes of the
 * 128bpp class stay below 2^24 so its 0x00ffffff masks never bite.
 *
 * Workgroup = 64 x 16 threads like k_pow2. For images narrower than 64 thread
 * columns (thumbnails) a wave folds over several output rows, so all 64 lanes
 * work: lane = (row within wave) * lpr + thread column, lpr a power of two. */

#pragma once

namespace smolhip {

template <int H, int IN>
struct Pow2xCfg
{
    static constexpr bool BPP3 = (IN == PX_N24 || IN == PX_W24);
    static constexpr bool WIDE = (IN == PX_W32 || IN == PX_W24 || IN == PX_W16);
    static constexpr int R = 2 << H;                              /* source px per output px, per axis */
    static constexpr int PXL = R >= 4 ? R : 4;                    /* source px per lane per row        */
    static constexpr int OPT = PXL / R;                           /* output px per lane                */
    static constexpr int NLOAD = PXL / 4;                         /* 4-pixel loads per lane per row    */
    static constexpr int LOADB = BPP3 ? 12 : 16;                  /* bytes per load                    */
    static constexpr int RCH = R;                                 /* source rows loaded per phase      */
};

/* one lane's load unit: 4 pixels = 16 bytes (32-bit pixels) or 12 bytes (24-bit) */
typedef uint32_t u32x3_native __attribute__ ((ext_vector_type (3)));
typedef u32x3_native u32x3a __attribute__ ((aligned (4)));

/* (4-byte alignment is all global_load_dwordx4 / dwordx3 need: rows of a cropped image keep these kernels) */
typedef u32x4 u32x4a __attribute__ ((aligned (4)));
template <bool BPP3> struct Pow2xRaw { typedef u32x4a T; };
template <> struct Pow2xRaw<true> { typedef u32x3a T; };

template <bool BPP3>
__device__ __forceinline__ typename Pow2xRaw<BPP3>::T pow2x_load (const gbyte *p)
{
    return __builtin_nontemporal_load (reinterpret_cast<const SMOL_AS_GLOBAL typename Pow2xRaw<BPP3>::T *> (p));
}

/* one load unit -> 4 pixel words (bytes in input memory order) */
__device__ __forceinline__ void pow2x_unit (const u32x4 &raw, uint32_t *w)
{
    w [0] = raw.x;
    w [1] = raw.y;
    w [2] = raw.z;
    w [3] = raw.w;
}

__device__ __forceinline__ void pow2x_unit (const u32x3_native &raw, uint32_t *w)
{
    const uint32_t d0 = raw.x, d1 = raw.y, d2 = raw.z;
    w [0] = d0 | 0xff000000u;
    w [1] = __builtin_amdgcn_alignbyte (d1, d0, 3) | 0xff000000u;
    w [2] = __builtin_amdgcn_alignbyte (d2, d1, 2) | 0xff000000u;
    w [3] = (d2 >> 8) | 0xff000000u;
}

/* NLOAD load units of one row -> PXL pixel words */
template <typename T, int NLOAD, int PXL>
__device__ __forceinline__ void pow2x_words (const T (&raw) [NLOAD], uint32_t (&w) [PXL])
{
#pragma unroll
    for (int l = 0; l < NLOAD; l++)
        pow2x_unit (raw [l], &w [l * 4]);
}

/* OPT packed 24-bit pixels (low 3 bytes of each word) -> memory at p */
template <int OPT>
__device__ __forceinline__ void pow2x_store24 (gbyte *p, const uint32_t (&w) [OPT])
{
    if (OPT >= 4)
    {
        SMOL_AS_GLOBAL uint32_t *d3 = reinterpret_cast<SMOL_AS_GLOBAL uint32_t *> (p);
#pragma unroll
        for (int g = 0; g < OPT / 4; g++)
        {
            d3 [g * 3 + 0] = w [g * 4 + 0] | (w [g * 4 + 1] << 24);
            d3 [g * 3 + 1] = (w [g * 4 + 1] >> 8) | (w [g * 4 + 2] << 16);
            d3 [g * 3 + 2] = (w [g * 4 + 2] >> 16) | (w [g * 4 + 3] << 8);
        }
    }
    else if (OPT == 2)
    {
        SMOL_AS_GLOBAL uint16_t *d2 = reinterpret_cast<SMOL_AS_GLOBAL uint16_t *> (p);
        d2 [0] = (uint16_t) w [0];
        d2 [1] = (uint16_t) ((w [0] >> 16) | (w [OPT - 1] << 8));
        d2 [2] = (uint16_t) (w [OPT - 1] >> 8);
    }
    else
    {
        p [0] = (uint8_t) w [0];
        p [1] = (uint8_t) (w [0] >> 8);
        p [2] = (uint8_t) (w [0] >> 16);
    }
}

/* unassociated word -> premultiplied word in packed 16-bit lanes:
 *   ((c + 1) * (a + 1) - 1) >> 8  ==  (c * (a + 1) + a) >> 8,  at most 65535 before the shift
 * (smolscale-generic.c:238-244); the alpha byte passes through. */
__device__ __forceinline__ uint32_t pow2x_premul_word (uint32_t w, uint32_t ain, uint32_t amask)
{
    const uint32_t a = (w >> (8 * ain)) & 0xffu;
    const u16x2 A2 = as_u16x2 (__umul24 (a, 0x00010001u));
    const u16x2 ONE2 = { 1, 1 };
    const u16x2 M2 = A2 + ONE2;
    const uint32_t lo = as_u32 ((as_u16x2 (w & PAIR_MASK) * M2 + A2) >> 8);
    const uint32_t hi = as_u32 ((as_u16x2 ((w >> 8) & PAIR_MASK) * M2 + A2) >> 8);
    const uint32_t r = lo | (hi << 8);
    return (r & ~amask) | (w & amask);
}

/* 128bpp-class accumulation of one lane's source block: acc [o][ch] = sum over the
 * 2^H vertical taps (to be shifted right by H) */
template <int H, int IN, int UK>
__device__ __forceinline__ void pow2x_wide_acc (const JobDev &J, const gbyte *sp, const LutLds &lut,
                                                uint32_t (&acc) [Pow2xCfg<H, IN>::OPT][4])
{
    using C = Pow2xCfg<H, IN>;
    const WideUnpack<UK> unpack (J, lut);
    const WideUnpack<WU_LIN_O> unpack_opaque (J, lut);

#pragma unroll
    for (int o = 0; o < C::OPT; o++)
#pragma unroll
        for (int i = 0; i < 4; i++)
            acc [o][i] = 0;

    /* One vertical tap = one pair of source rows. Rows come in chunks of up to four, every load of a chunk issued
     * before the first conversion (a rolled pair loop with a "load the next pair" branch at its head ended up with
     * those loads at the loop's tail, right in front of their wait: no overlap at all). */
    constexpr int CH = C::R < 4 ? C::R : 4;
#pragma unroll 1
    for (int r0 = 0; r0 < C::R; r0 += CH)
    {
        typename Pow2xRaw<C::BPP3>::T raw [CH][C::NLOAD];
#pragma unroll
        for (int r = 0; r < CH; r++)
#pragma unroll
            for (int l = 0; l < C::NLOAD; l++)
                raw [r][l] = pow2x_load<C::BPP3> (sp + (size_t) (r0 + r) * J.src_stride + l * C::LOADB);

#pragma unroll
      for (int rp = 0; rp < CH / 2; rp++)
      {
        uint32_t wa [C::PXL], wb [C::PXL];
        pow2x_words (raw [2 * rp], wa);
        pow2x_words (raw [2 * rp + 1], wb);

        /* photographs with an alpha channel: when every pixel of this row pair is opaque on the whole wave the
         * conversion is one fused look-up per channel */
        bool opaque = false;
        if (UK == WU_LIN_P || UK == WU_LIN_U)
        {
            uint32_t all = wa [0] & wb [0];
#pragma unroll
            for (int i = 1; i < C::PXL; i++)
                all &= wa [i] & wb [i];
            opaque = wide_batch_opaque (J, all);
        }

        auto taps = [&] (const auto &un)
        {
#pragma unroll
            for (int o = 0; o < C::OPT; o++)
            {
                /* horizontal, per row: 2^H taps of two pixels each, then the halving shift */
                uint32_t ha [4] = { 0, 0, 0, 0 }, hb [4] = { 0, 0, 0, 0 };
#pragma unroll
                for (int t = 0; t < (1 << H); t++)
                {
                    uint32_t p [4], q [4];
                    un (wa [o * C::R + 2 * t], p);
                    un (wa [o * C::R + 2 * t + 1], q);
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        ha [i] += (p [i] + q [i]) >> 1;
                    un (wb [o * C::R + 2 * t], p);
                    un (wb [o * C::R + 2 * t + 1], q);
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        hb [i] += (p [i] + q [i]) >> 1;
                }
                /* vertical tap of the two rows' results */
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc [o][i] += ((ha [i] >> H) + (hb [i] >> H)) >> 1;
            }
        };
        if (opaque)
            taps (unpack_opaque);
        else
            taps (unpack);
      }
    }
}

template <int H, int IN>
__device__ __forceinline__ void pow2x_body (const JobDev &J)
{
    using C = Pow2xCfg<H, IN>;

    const uint32_t lane = threadIdx.x;
    const uint32_t shift = J.px_lpr_shift;                 /* log2 (lanes per output row in a wave) */
    const uint32_t lpr = 1u << shift;
    const uint32_t tcols = J.w_out / C::OPT;
    const uint32_t bx = blockIdx.x, by = blockIdx.y;       /* grid = (column tiles, row tiles, jobs) */
    const uint32_t row = ((by * 16 + threadIdx.y) << (6 - shift)) + (lane >> shift);
    const uint32_t tc = (bx << shift) + (lane & (lpr - 1));
    const bool live = row < J.n_rows && tc < tcols;

#ifdef SMOL_KERNEL_EMU
    static uint8_t lut_lds [LUT_LDS_BYTES];
    if (C::WIDE)
        lut_fill_lds (lut_lds, 0, 1);                      /* emulated lanes run one after the other */
    if (!live)
        return;
#else
    __shared__ __attribute__ ((aligned (16))) uint8_t lut_lds [C::WIDE ? LUT_LDS_BYTES : 16];
    if (C::WIDE)
    {
        lut_fill_lds (lut_lds, threadIdx.y * 64 + lane, 1024);
        __syncthreads ();
    }
    if (!live)
        return;
#endif

    const uint32_t y = J.first_row + row;
    const gbyte *sp = gptr (J.src) + (size_t) y * C::R * J.src_stride + (size_t) tc * (C::NLOAD * C::LOADB);
    gbyte *drow = gptr (J.dst) + (size_t) row * J.dst_stride;

    if (!C::WIDE)
    {
        const uint32_t ain = J.pos_in [3];
        const uint32_t amask = 0xffu << (8 * ain);
        const bool premul_in = (IN == PX_N32) && J.unassoc_in;
        uint32_t hres [C::OPT][C::R];

#pragma unroll
        for (int r0 = 0; r0 < C::R; r0 += C::RCH)
        {
            typename Pow2xRaw<C::BPP3>::T raw [C::RCH][C::NLOAD];
#pragma unroll
            for (int r = 0; r < C::RCH; r++)
#pragma unroll
                for (int l = 0; l < C::NLOAD; l++)
                    raw [r][l] = pow2x_load<C::BPP3> (sp + (size_t) (r0 + r) * J.src_stride + l * C::LOADB);
#pragma unroll
            for (int r = 0; r < C::RCH; r++)
            {
                uint32_t w [C::PXL];
                pow2x_words (raw [r], w);
                if (premul_in)
                {
#pragma unroll
                    for (int i = 0; i < C::PXL; i++)
                        w [i] = pow2x_premul_word (w [i], ain, amask);
                }
#pragma unroll
                for (int o = 0; o < C::OPT; o++)
                    hres [o][r0 + r] = reduce_pow2<H> (&w [o * C::R]);
            }
        }

        uint32_t outw [C::OPT];
#pragma unroll
        for (int o = 0; o < C::OPT; o++)
            outw [o] = tile_pack<OUT_ANY> (J, px2_from_word (reduce_pow2<H> (hres [o])));

        if (J.bpp_out == 4)
        {
            gbyte *dp = drow + (size_t) tc * (C::OPT * 4);
            if (C::OPT == 1)
                store_out32 (reinterpret_cast<SMOL_AS_GLOBAL uint32_t *> (dp), outw [0]);
            else if (C::OPT == 2)
            {
                store_out64 (reinterpret_cast<SMOL_AS_GLOBAL uint64_t *> (dp), outw [0], outw [C::OPT - 1]);
            }
            else
            {
#pragma unroll
                for (int g = 0; g < C::OPT / 4; g++)
                {
                    const u32x4 v4 = { outw [(g * 4 + 0) % C::OPT], outw [(g * 4 + 1) % C::OPT],
                                       outw [(g * 4 + 2) % C::OPT], outw [(g * 4 + 3) % C::OPT] };
                    store_out128 (dp + (size_t) g * 16, v4);
                }
            }
        }
        else
        {
#pragma unroll
            for (int o = 0; o < C::OPT; o++)
                outw [o] &= 0x00ffffffu;
            pow2x_store24<C::OPT> (drow + (size_t) tc * (C::OPT * 3), outw);
        }
    }
    else
    {
        const LutLds lut (lut_lds);
        uint32_t acc [C::OPT][4];

        /* the conversions' flavour is fixed here, once per lane (smol_kernels.hip.h WideUnpack / WidePack) */
        const int uk = wide_unpack_kind (J);
        if (IN == PX_W16)
            pow2x_wide_acc<H, IN, WU_P16> (J, sp, lut, acc);                 /* (pow2x_setup: premul16 without sRGB) */
        else if (IN == PX_W24)
        {
            if (uk == WU_LIN_O)
                pow2x_wide_acc<H, IN, WU_LIN_O> (J, sp, lut, acc);
            else
                pow2x_wide_acc<H, IN, WU_ANY> (J, sp, lut, acc);
        }
        else
        {
            switch (uk)
            {
                case WU_P16L:  pow2x_wide_acc<H, IN, WU_P16L> (J, sp, lut, acc); break;
                case WU_LIN_P: pow2x_wide_acc<H, IN, WU_LIN_P> (J, sp, lut, acc); break;
                case WU_LIN_U: pow2x_wide_acc<H, IN, WU_LIN_U> (J, sp, lut, acc); break;
                default:       pow2x_wide_acc<H, IN, WU_ANY> (J, sp, lut, acc); break;
            }
        }

        auto store = [&] (const auto &pack)
        {
#pragma unroll
            for (int o = 0; o < C::OPT; o++)
            {
                uint32_t v [4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    v [i] = acc [o][i] >> H;
                wide_store_word (J, drow, tc * C::OPT + o, pack (v));
            }
        };
        if (IN == PX_W16)
            store (WidePack<WP_P16> (J, lut));
        else switch (wide_pack_kind (J))
        {
            case WP_P16L:  store (WidePack<WP_P16L> (J, lut)); break;
            case WP_LIN_P: store (WidePack<WP_LIN_P> (J, lut)); break;
            case WP_LIN_U: store (WidePack<WP_LIN_U> (J, lut)); break;
            case WP_LIN_Q: store (WidePack<WP_LIN_Q> (J, lut)); break;
            default:       store (WidePack<WP_ANY> (J, lut)); break;
        }
    }
}

/* 2:1 and 4:1 hold two 1024-thread workgroups per CU (8 waves per SIMD) only up to 64 vector registers: say so
 * (the specialised 128bpp conversions would otherwise take 76 and halve the occupancy: C5mix 468 -> 500 us) */
#ifdef SMOL_KERNEL_EMU
#define SMOL_POW2X_OCC(H)
#else
#define SMOL_POW2X_OCC(H) __attribute__ ((amdgpu_waves_per_eu ((H) <= 1 ? 8 : 4)))
#endif

template <int H, int IN>
__global__ void __launch_bounds__ (1024) SMOL_POW2X_OCC (H)
k_pow2x_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.z);         /* uniform: scalar loads, once */
    pow2x_body<H, IN> (J);
}

/* A batch that mixes pixel types (BASELINE config 5 as specified: 10 000 thumbnails, types drawn from the
 * whole table) in ONE launch: workgroups walk the images in the caller's order -- which is address order
 * for an arena -- and each switches, wave-uniformly, to its job's flavour. Four launches, one per flavour,
 * each skipping through the arena, lose 8-10 % to their ramps and tails (profiles/r02u_pow2x_types.txt);
 * running them side by side on forked streams is slower still. */
template <int H>
__global__ void __launch_bounds__ (1024) SMOL_POW2X_OCC (H)
k_pow2x_mixed_batch (const JobDev *__restrict__ jobs)
{
    const JobDev J = load_job (jobs, blockIdx.z);
    switch (J.px_kind)
    {
        case PX_N32: pow2x_body<H, PX_N32> (J); break;
        case PX_N24: pow2x_body<H, PX_N24> (J); break;
        case PX_W32: pow2x_body<H, PX_W32> (J); break;
        case PX_W16: pow2x_body<H, PX_W16> (J); break;
        default:     pow2x_body<H, PX_W24> (J); break;
    }
}

template <int H, int IN>
__global__ void __launch_bounds__ (1024) SMOL_POW2X_OCC (H)
k_pow2x_one (const JobDev J)
{
    pow2x_body<H, IN> (J);
}

} /* namespace smolhip */
