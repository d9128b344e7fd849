/* smol_cpu.c -- the library's CPU backend: the whole resampling path on the
 * host cores, from scratch, behind the same seam as the HIP backend (smolscale_api.c). Never the
 * default: it runs only when the caller asks for it (SMOL_BACKEND=cpu, or SMOL_BACKEND=auto on a
 * machine without a usable GPU); with the default SMOL_BACKEND=hip a missing GPU is still a loud
 * abort().
 *
 * What it has to reproduce is the reference's integer fixed-point path; the arithmetic is restated
 * per channel exactly as the HIP kernels restate it (smol_kernels.hip.h: unpack_word / hsample /
 * general_pixel / store_px), so that the two backends of this library agree byte for byte with each
 * other and with the reference's GENERIC implementation:
 *   unpack    smolscale-generic.c:353-464 (64bpp class), :470-752 (128bpp class)
 *   horizontal :1290-1611, vertical :1648-2318, pack :758-1164
 * Known reference inconsistency, decided explicitly: for unassociated -> unassociated (premul16) the
 * reference's generic packer WRAPS (& 0xff, smolscale-generic.c:290-299) where its AVX2 packer
 * SATURATES (smolscale-avx2.c:1045-1047). This backend wraps, like the generic code, the oracle and
 * the HIP kernels.
 *
 * Shape: rows. A source row is unpacked to four lanes per pixel, filtered horizontally with
 * per-column tap tables built once per call from the closed-form sampling positions, and the
 * horizontally filtered rows meet in the vertical filter through a small row cache (bilinear) or an
 * accumulator (box). Three code paths:
 *   - generic: 32-bit lanes, every job (both storage classes, every pixel type, sRGB) -- what runs
 *     without AVX2, under SMOL_CPU_SIMD=0 and on the last pixels of a row;
 *   - fast_*: the 64bpp class (every pixel type and filter) in packed 16-bit lanes with AVX2 /
 *     AVX-512BW intrinsics, selected at run time (__builtin_cpu_supports);
 *   - wide_*: the 128bpp class in 32-bit lanes with AVX2 (round 5; rounds 3-4 had SIMD for 32-bit
 *     premultiplied pixels under a bilinear filter on both axes only). */

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined (__x86_64__)
#include <immintrin.h>
#define SMOL_CPU_X86 1
/* the generic (32-bit-lane) row functions are compiled twice, the loader picks the AVX2 copy where the CPU has it: their
 * loops are plain arithmetic over lanes that the compiler vectorises itself */
#define SMOL_CPU_CLONES __attribute__ ((target_clones ("avx2", "default")))
#else
#define SMOL_CPU_CLONES
#endif

#include "smol_hip.h"
#include "smol_cpu.h"
#include "smol_luts.h"

enum { F_COPY = SMOL_HIP_FILTER_COPY, F_ONE = SMOL_HIP_FILTER_ONE, F_BILINEAR = SMOL_HIP_FILTER_BILINEAR, F_BOX = SMOL_HIP_FILTER_BOX };

/* byte position of R,G,B,A per SmolPixelType (smolscale.h:16-36 names the memory order) */
typedef struct { uint8_t bpp; uint8_t pos [4]; uint8_t unassoc; } TypeInfo;

static const TypeInfo k_types [10] = {
    { 4, { 0, 1, 2, 3 }, 0 }, { 4, { 2, 1, 0, 3 }, 0 }, { 4, { 1, 2, 3, 0 }, 0 }, { 4, { 3, 2, 1, 0 }, 0 },
    { 4, { 0, 1, 2, 3 }, 1 }, { 4, { 2, 1, 0, 3 }, 1 }, { 4, { 1, 2, 3, 0 }, 1 }, { 4, { 3, 2, 1, 0 }, 1 },
    { 3, { 0, 1, 2, 0xff }, 0 }, { 3, { 2, 1, 0, 0xff }, 0 },
};

typedef struct
{
    const SmolHipJob *job;
    const TypeInfo *ti, *to;
    int wide, p16, linear, quirk24;
    uint32_t mask;                      /* lane mask of the storage class: 0xff / 0x00ffffff */
    /* horizontal tables */
    uint32_t *h_ofs;                    /* bilinear: source index of sample j; box: first index of box x */
    uint16_t *h_f;                      /* bilinear: weight F of sample j (0..256)                        */
    uint32_t *h_last;                   /* box: last index of box x                                       */
    uint16_t *h_wl, *h_fr;              /* box: left / right edge weights                                 */
}
Ctx;

/* ---- per-channel arithmetic (the formulas of smol_kernels.hip.h) ------------------------------- */

static inline uint32_t premul8 (uint32_t c, uint32_t a) { return (((c + 1u) * (a + 1u) - 1u) >> 8) & 0xffu; }
static inline uint32_t premul8l (uint32_t l, uint32_t a) { return (((l + 1u) * ((a << 3) + 1u) - 1u) >> 11) & 0x7ffu; }
static inline uint32_t unpremul8 (uint32_t c, uint32_t a) { return (uint32_t) ((((uint64_t) c * smol_inv_p8_lut [a]) >> 13) & 0xffu); }

static inline uint32_t
lerp (uint32_t p, uint32_t q, uint32_t F, uint32_t mask)
{
    int32_t d = ((int32_t) p - (int32_t) q) * (int32_t) F;
    return (uint32_t) ((d >> 8) + (int32_t) q) & mask;
}

static inline uint32_t
box_scale (const Ctx *c, uint32_t acc, uint32_t mul)
{
    uint64_t v;
    if (!c->wide)
        acc &= 0xffffu;
    v = (uint64_t) acc * mul + (1u << 23);
    return (uint32_t) (v >> 24) & (c->wide ? 0xffffu : 0xffu);
}

static void
bilin_tap (uint64_t pos, uint32_t dim_in, uint32_t *ofs, uint32_t *F)
{
    uint32_t o = (uint32_t) (pos >> 32);
    if (o >= dim_in - 1) { *ofs = dim_in - 2; *F = 0; }
    else { *ofs = o; *F = 256u - ((uint32_t) (pos >> 24) & 0xffu); }
}

static void
box_span (uint64_t step, uint32_t dim_in, uint32_t k, uint32_t *o0, uint32_t *o1, uint32_t *f_right, uint32_t *w_left)
{
    uint64_t b = (uint64_t) k * step, e = b + step;
    uint32_t last = (uint32_t) (e >> 16);
    *o0 = (uint32_t) (b >> 16);
    *o1 = last > dim_in - 1 ? dim_in - 1 : last;
    *f_right = (uint32_t) (e >> 8) & 0xffu;
    *w_left = k ? 255u - ((uint32_t) (b >> 8) & 0xffu) : 256u;
}

/* ---- unpack one row: memory pixels -> four 32-bit lanes (R, G, B, alpha lane) -------------------- */

SMOL_CPU_CLONES
static void
unpack_row (const Ctx *c, const uint8_t *src, uint32_t n, uint32_t *out)
{
    const TypeInfo *ti = c->ti;
    const uint32_t bpp = ti->bpp, pr = ti->pos [0], pg = ti->pos [1], pb = ti->pos [2], pa = ti->pos [3];
    uint32_t i, k;

    for (i = 0; i < n; i++, src += bpp, out += 4)
    {
        uint32_t ch [3] = { src [pr], src [pg], src [pb] };
        const uint32_t a = (bpp == 4) ? src [pa] : 0xffu;

        if (c->wide && c->p16)
        {
            for (k = 0; k < 3; k++)
                out [k] = (c->linear ? smol_from_srgb_lut [ch [k]] : ch [k]) * a;
            out [3] = (a << 8) | 0x80u;
            continue;
        }
        if (c->wide && c->linear)
        {
            for (k = 0; k < 3; k++)
            {
                uint32_t v = ch [k];
                if (bpp == 4 && !ti->unassoc)
                    v = unpremul8 (v, a);
                else if (bpp == 3)
                    v = unpremul8 (v, 0xffu);
                out [k] = premul8l (smol_from_srgb_lut [v], a);
            }
        }
        else if (ti->unassoc)
        {
            for (k = 0; k < 3; k++)
                out [k] = premul8 (ch [k], a);
        }
        else
        {
            out [0] = ch [0]; out [1] = ch [1]; out [2] = ch [2];
        }
        out [3] = a;
    }
}

/* ---- pack one row ---------------------------------------------------------------------------- */

SMOL_CPU_CLONES
static void
pack_row (const Ctx *c, const uint32_t *in, uint32_t n, uint8_t *dst)
{
    const TypeInfo *to = c->to;
    const uint32_t bpp = to->bpp;
    uint32_t i, k;

    for (i = 0; i < n; i++, in += 4, dst += bpp)
    {
        uint32_t a, o [3];

        a = (c->wide && c->p16) ? (in [3] >> 8) & 0xffu : in [3] & 0xffu;
        for (k = 0; k < 3; k++)
        {
            const uint32_t s = in [k];

            if (!c->wide || (!c->p16 && !c->linear))
                o [k] = to->unassoc ? unpremul8 (s, a) : (s & 0xffu);
            else if (c->p16)
            {
                if (c->linear)
                    o [k] = smol_to_srgb_lut [(uint32_t) (((uint64_t) s * smol_inv_p16l_lut [a]) >> 19) & 0x7ffu];
                else
                    o [k] = (uint32_t) (((uint64_t) s * smol_inv_p16_lut [a]) >> 16) & 0xffu;    /* wraps: see the header */
            }
            else
            {
                const uint32_t u = (uint32_t) (((uint64_t) s * smol_inv_p8l_lut [a]) >> 10) & 0x7ffu;

                if (bpp == 3)
                    o [k] = c->quirk24 ? smol_to_srgb_lut [s & 0x7ffu] : smol_to_srgb_lut [u];
                else if (to->unassoc)
                    o [k] = smol_to_srgb_lut [u];
                else
                    o [k] = premul8 (smol_to_srgb_lut [u], a);
            }
        }
        dst [to->pos [0]] = (uint8_t) o [0];
        dst [to->pos [1]] = (uint8_t) o [1];
        dst [to->pos [2]] = (uint8_t) o [2];
        if (bpp == 4)
            dst [to->pos [3]] = (uint8_t) a;
    }
}

/* ---- horizontal filter of one unpacked row ------------------------------------------------------ */

SMOL_CPU_CLONES
static void
hfilter_row (const Ctx *c, const uint32_t *u, uint32_t *out)
{
    const SmolHipJob *j = c->job;
    const uint32_t w_out = j->width_out, mask = c->mask;
    uint32_t x, k;

    if (j->filter_h == F_BILINEAR)
    {
        const uint32_t h = j->halvings_h, ns = 1u << h;
        const uint32_t *ofs = c->h_ofs;
        const uint16_t *f = c->h_f;

        for (x = 0; x < w_out; x++, out += 4)
        {
            uint32_t acc [4] = { 0, 0, 0, 0 }, s;

            for (s = 0; s < ns; s++, ofs++, f++)
            {
                const uint32_t *p = u + (size_t) *ofs * 4, *q = p + 4;
                for (k = 0; k < 4; k++)
                    acc [k] += lerp (p [k], q [k], *f, mask);
            }
            for (k = 0; k < 4; k++)
                out [k] = (acc [k] >> h) & mask;
        }
    }
    else if (j->filter_h == F_BOX)
    {
        for (x = 0; x < w_out; x++, out += 4)
        {
            const uint32_t o0 = c->h_ofs [x], o1 = c->h_last [x], wl = c->h_wl [x], fr = c->h_fr [x];
            uint32_t acc [4] = { 0, 0, 0, 0 }, s;

            for (s = o0; s <= o1; s++)
            {
                const uint32_t *p = u + (size_t) s * 4;
                if (s == o0)
                    for (k = 0; k < 4; k++) acc [k] += ((p [k] * wl) >> 8) & mask;
                else if (s == o1)
                    for (k = 0; k < 4; k++) acc [k] += ((p [k] * fr) >> 8) & mask;
                else
                    for (k = 0; k < 4; k++) acc [k] += p [k];
            }
            for (k = 0; k < 4; k++)
                out [k] = box_scale (c, acc [k], j->span_mul_x);
        }
    }
    else if (j->filter_h == F_ONE)
    {
        for (x = 0; x < w_out; x++, out += 4)
            memcpy (out, u, 16);
    }
    else
        memcpy (out, u, (size_t) w_out * 16);
}

/* ---- horizontally filtered rows on demand -------------------------------------------------------- */

typedef struct
{
    const Ctx *c;
    const uint8_t *src;
    uint32_t stride;
    uint32_t *urow;                     /* one unpacked source row                     */
    uint32_t *hrow [3];                 /* cache of horizontally filtered rows         */
    uint32_t tag [3];                   /* source row each slot holds (UINT32_MAX: none) */
    uint32_t next;
}
RowSource;

/* `keep': a row handed out a moment ago that must stay valid (NULL: none) */
static const uint32_t *
hrow_get (RowSource *rs, uint32_t r, const uint32_t *keep)
{
    uint32_t i;

    for (i = 0; i < 3; i++)
        if (rs->tag [i] == r)
            return rs->hrow [i];
    i = rs->next;
    if (rs->hrow [i] == keep)
        i = (i + 1) % 3;
    rs->next = (i + 1) % 3;
    unpack_row (rs->c, rs->src + (size_t) r * rs->stride, rs->c->job->width_in, rs->urow);
    hfilter_row (rs->c, rs->urow, rs->hrow [i]);
    rs->tag [i] = r;
    return rs->hrow [i];
}

/* ---- the fast class: 32-bit premultiplied in and out, bilinear on both axes, 64bpp storage --------- */

#ifdef SMOL_CPU_X86

/* one source pixel as a 32-bit word in memory byte order; 24-bit pixels get alpha = 255 as byte 3 */
static inline __attribute__ ((always_inline)) uint32_t
fast_px (const uint8_t *row, uint32_t o, const uint32_t bpp)
{
    uint32_t w;
    uint16_t lo;
    if (bpp == 4)
    {
        memcpy (&w, row + (size_t) o * 4, 4);
        return w;
    }
    row += (size_t) o * 3;
    memcpy (&lo, row, 2);
    return (uint32_t) lo | ((uint32_t) row [2] << 16) | 0xff000000u;
}

/* horizontal: one source row of 3- or 4-byte pixels -> w_out pixels of four 16-bit lanes (memory byte order; a 24-bit
 * pixel's fourth lane is 255). Unassociated pixels are premultiplied on the way in, as the reference's unpackers do
 * (smolscale-generic.c:238-244, 415-464): ((c + 1) * (a + 1) - 1) >> 8 in 16-bit lanes -- (c + 1) * (a + 1) reaches 2^16
 * only as 256 * 256, which wraps to 0, and the - 1 then gives 0xffff: the same value modulo 2^16. */
__attribute__ ((target ("avx2"), always_inline))
static inline void
fast_hrow_body (const Ctx *c, const uint8_t *src, uint16_t *out, const uint32_t bpp, const int pm)
{
    const SmolHipJob *j = c->job;
    const uint32_t w_out = j->width_out, h = j->halvings_h, ns = 1u << h;
    const uint32_t apos = c->ti->pos [3] & 3u;
    const uint32_t *ofs = c->h_ofs;
    const uint16_t *f = c->h_f;
    const __m128i shift = _mm_cvtsi32_si128 ((int) h);
    uint8_t bsel [32];
    uint16_t amask [16];
    __m256i v_bsel, v_amask;
    const __m256i one = _mm256_set1_epi16 (1);
    uint32_t x = 0, i;

    /* per pixel (four 16-bit lanes = 8 bytes): every lane <- the alpha lane; and a mask that is all ones on alpha lanes */
    for (i = 0; i < 32; i++)
        bsel [i] = (uint8_t) ((i & 8u) + 2u * apos + (i & 1u));
    for (i = 0; i < 16; i++)
        amask [i] = (i & 3u) == apos ? 0xffffu : 0u;
    v_bsel = _mm256_loadu_si256 ((const __m256i *) bsel);
    v_amask = _mm256_loadu_si256 ((const __m256i *) amask);

    /* four output pixels per iteration: 16 lanes of 16 bits */
    for (; x + 4 <= w_out; x += 4, out += 16)
    {
        __m256i acc = _mm256_setzero_si256 ();
        uint32_t s;

        for (s = 0; s < ns; s++)
        {
            uint32_t p [4], q [4];
            uint64_t fw [4];
            __m256i vp, vq, vf, vg;

            for (i = 0; i < 4; i++)
            {
                const uint32_t o = ofs [(x + i) * ns + s];
                p [i] = fast_px (src, o, bpp);
                q [i] = fast_px (src, o + 1, bpp);
                fw [i] = (uint64_t) f [(x + i) * ns + s] * 0x0001000100010001ull;
            }
            vp = _mm256_cvtepu8_epi16 (_mm_loadu_si128 ((const __m128i *) p));
            vq = _mm256_cvtepu8_epi16 (_mm_loadu_si128 ((const __m128i *) q));
            if (pm)
            {
                const __m256i mp = _mm256_add_epi16 (_mm256_shuffle_epi8 (vp, v_bsel), one);
                const __m256i mq = _mm256_add_epi16 (_mm256_shuffle_epi8 (vq, v_bsel), one);
                const __m256i rp = _mm256_srli_epi16 (_mm256_sub_epi16 (_mm256_mullo_epi16 (_mm256_add_epi16 (vp, one), mp), one), 8);
                const __m256i rq = _mm256_srli_epi16 (_mm256_sub_epi16 (_mm256_mullo_epi16 (_mm256_add_epi16 (vq, one), mq), one), 8);
                vp = _mm256_blendv_epi8 (rp, vp, v_amask);
                vq = _mm256_blendv_epi8 (rq, vq, v_amask);
            }
            vf = _mm256_loadu_si256 ((const __m256i *) fw);
            vg = _mm256_sub_epi16 (_mm256_set1_epi16 (256), vf);
            /* (p*F + q*(256-F)) >> 8 == q + (((p-q)*F) >> 8) for 8-bit lanes: at most 255*256, fits 16 bits */
            acc = _mm256_add_epi16 (acc, _mm256_srli_epi16 (_mm256_add_epi16 (_mm256_mullo_epi16 (vp, vf),
                                                                              _mm256_mullo_epi16 (vq, vg)), 8));
        }
        _mm256_storeu_si256 ((__m256i *) out, _mm256_srl_epi16 (acc, shift));
    }
    for (; x < w_out; x++, out += 4)
    {
        uint32_t acc [4] = { 0, 0, 0, 0 }, s, k;
        for (s = 0; s < ns; s++)
        {
            const uint32_t o = ofs [x * ns + s];
            const uint32_t wp = fast_px (src, o, bpp), wq = fast_px (src, o + 1, bpp);
            const uint32_t F = f [x * ns + s];
            for (k = 0; k < 4; k++)
            {
                uint32_t pv = (wp >> (8 * k)) & 0xffu, qv = (wq >> (8 * k)) & 0xffu;
                if (pm && k != apos)
                {
                    pv = premul8 (pv, (wp >> (8 * apos)) & 0xffu);
                    qv = premul8 (qv, (wq >> (8 * apos)) & 0xffu);
                }
                acc [k] += lerp (pv, qv, F, 0xffu);
            }
        }
        for (k = 0; k < 4; k++)
            out [k] = (uint16_t) (acc [k] >> h);
    }
}

/* (one specialisation per source kind: the pixel fetch and the premultiply are compile-time choices in the loop) */
__attribute__ ((target ("avx2")))
static void
fast_hrow_avx2 (const Ctx *c, const uint8_t *src, uint16_t *out)
{
    if (c->ti->bpp == 3)
        fast_hrow_body (c, src, out, 3, 0);
    else if (c->ti->unassoc)
        fast_hrow_body (c, src, out, 4, 1);
    else
        fast_hrow_body (c, src, out, 4, 0);
}

/* vertical: acc += (top*F + bot*(256-F)) >> 8 over n 16-bit lanes */
__attribute__ ((target ("avx2")))
static void
fast_vblend_avx2 (const uint16_t *top, const uint16_t *bot, uint32_t F, uint16_t *acc, int first, size_t n)
{
    const __m256i vf = _mm256_set1_epi16 ((short) F), vg = _mm256_set1_epi16 ((short) (256 - F));
    size_t i = 0;

    for (; i + 16 <= n; i += 16)
    {
        const __m256i t = _mm256_loadu_si256 ((const __m256i *) (top + i));
        const __m256i b = _mm256_loadu_si256 ((const __m256i *) (bot + i));
        __m256i v = _mm256_srli_epi16 (_mm256_add_epi16 (_mm256_mullo_epi16 (t, vf), _mm256_mullo_epi16 (b, vg)), 8);
        if (!first)
            v = _mm256_add_epi16 (v, _mm256_loadu_si256 ((const __m256i *) (acc + i)));
        _mm256_storeu_si256 ((__m256i *) (acc + i), v);
    }
    for (; i < n; i++)
    {
        const uint16_t v = (uint16_t) ((top [i] * F + bot [i] * (256 - F)) >> 8);
        acc [i] = first ? v : (uint16_t) (acc [i] + v);
    }
}

__attribute__ ((target ("avx512bw")))
static void
fast_vblend_avx512 (const uint16_t *top, const uint16_t *bot, uint32_t F, uint16_t *acc, int first, size_t n)
{
    const __m512i vf = _mm512_set1_epi16 ((short) F), vg = _mm512_set1_epi16 ((short) (256 - F));
    size_t i = 0;

    for (; i + 32 <= n; i += 32)
    {
        const __m512i t = _mm512_loadu_si512 ((const void *) (top + i));
        const __m512i b = _mm512_loadu_si512 ((const void *) (bot + i));
        __m512i v = _mm512_srli_epi16 (_mm512_add_epi16 (_mm512_mullo_epi16 (t, vf), _mm512_mullo_epi16 (b, vg)), 8);
        if (!first)
            v = _mm512_add_epi16 (v, _mm512_loadu_si512 ((const void *) (acc + i)));
        _mm512_storeu_si512 ((void *) (acc + i), v);
    }
    for (; i < n; i++)
    {
        const uint16_t v = (uint16_t) ((top [i] * F + bot [i] * (256 - F)) >> 8);
        acc [i] = first ? v : (uint16_t) (acc [i] + v);
    }
}

/* (acc >> hv) narrowed to bytes and permuted from input to output channel order. perm [k]: the input lane that holds
 * output byte k's channel. Unassociated output: the colour lanes are un-premultiplied through the inverse table
 * ((c * inv [a]) >> 13 & 0xff, smolscale-generic.c:227-236, 246-259) in 32-bit lanes, two pixels per vector. 24-bit
 * output: four-byte pixels compressed to three with one byte shuffle per 128-bit half. */
__attribute__ ((target ("avx2"), always_inline))
static inline void
fast_pack_body (const Ctx *c, const uint16_t *acc, uint32_t hv, const uint8_t perm [4], uint32_t n_px, uint8_t *dst,
                const uint32_t bpp_out, const int unassoc)
{
    const __m128i shift = _mm_cvtsi32_si128 ((int) hv);
    /* the input lane that carries alpha (24-bit sources: the synthetic fourth lane) */
    const uint32_t alane = c->ti->bpp == 4 ? (c->ti->pos [3] & 3u) : 3u;
    uint8_t sel [32];
    uint32_t i, k;
    __m256i vsel;

    if (bpp_out == 4)
        for (i = 0; i < 32; i++)
            sel [i] = (uint8_t) ((i & ~3u) + perm [i & 3u]);
    else
        for (i = 0; i < 32; i++)
        {
            /* per 128-bit half: bytes 0 .. 11 <- the three colour bytes of its four pixels, the rest zero */
            const uint32_t b = i & 15u;
            sel [i] = b < 12u ? (uint8_t) (4u * (b / 3u) + perm [b % 3u]) : 0x80u;
        }
    vsel = _mm256_loadu_si256 ((const __m256i *) sel);

    i = 0;
    /* (24-bit output: the second half's 16-byte store reaches 4 bytes past its 12; keep those inside the row) */
    for (; i + 8 + (bpp_out == 3 ? 2u : 0u) <= n_px; i += 8)
    {
        __m256i a = _mm256_srl_epi16 (_mm256_loadu_si256 ((const __m256i *) (acc + (size_t) i * 4)), shift);
        __m256i b = _mm256_srl_epi16 (_mm256_loadu_si256 ((const __m256i *) (acc + (size_t) i * 4 + 16)), shift);
        __m256i p;

        if (unassoc)
        {
            /* 16 lanes of 16 bits = 4 pixels per vector -> 32-bit lanes, two pixels per vector */
            __m256i v [4], r [4];
            uint32_t u;
            const __m256i amask32 = _mm256_cmpeq_epi32 (_mm256_and_si256 (_mm256_set_epi32 (7, 6, 5, 4, 3, 2, 1, 0), _mm256_set1_epi32 (3)),
                                                        _mm256_set1_epi32 ((int) alane));
            v [0] = _mm256_cvtepu16_epi32 (_mm256_castsi256_si128 (a));
            v [1] = _mm256_cvtepu16_epi32 (_mm256_extracti128_si256 (a, 1));
            v [2] = _mm256_cvtepu16_epi32 (_mm256_castsi256_si128 (b));
            v [3] = _mm256_cvtepu16_epi32 (_mm256_extracti128_si256 (b, 1));
            for (u = 0; u < 4; u++)
            {
                /* every lane of a pixel <- its alpha lane, then the table */
                __m256i al;
                switch (alane)
                {
                    case 0:  al = _mm256_shuffle_epi32 (v [u], 0x00); break;
                    case 1:  al = _mm256_shuffle_epi32 (v [u], 0x55); break;
                    case 2:  al = _mm256_shuffle_epi32 (v [u], 0xaa); break;
                    default: al = _mm256_shuffle_epi32 (v [u], 0xff); break;
                }
                const __m256i inv = _mm256_i32gather_epi32 ((const int *) smol_inv_p8_lut, al, 4);
                const __m256i un = _mm256_and_si256 (_mm256_srli_epi32 (_mm256_mullo_epi32 (v [u], inv), 13), _mm256_set1_epi32 (0xff));
                r [u] = _mm256_blendv_epi8 (un, v [u], amask32);
            }
            /* back to 16-bit lanes in pixel order (packus works per 128-bit half) */
            a = _mm256_permute4x64_epi64 (_mm256_packus_epi32 (r [0], r [1]), 0xd8);
            b = _mm256_permute4x64_epi64 (_mm256_packus_epi32 (r [2], r [3]), 0xd8);
        }
        /* packus works per 128-bit half: restore the order, then permute bytes inside each pixel */
        p = _mm256_permute4x64_epi64 (_mm256_packus_epi16 (a, b), 0xd8);
        p = _mm256_shuffle_epi8 (p, vsel);
        if (bpp_out == 4)
            _mm256_storeu_si256 ((__m256i *) (dst + (size_t) i * 4), p);
        else
        {
            _mm_storeu_si128 ((__m128i *) (dst + (size_t) i * 3), _mm256_castsi256_si128 (p));
            _mm_storeu_si128 ((__m128i *) (dst + (size_t) i * 3 + 12), _mm256_extracti128_si256 (p, 1));
        }
    }
    for (; i < n_px; i++)
    {
        const uint32_t a8 = (uint32_t) (acc [(size_t) i * 4 + alane] >> hv);
        for (k = 0; k < bpp_out; k++)
        {
            uint32_t v = (uint32_t) (acc [(size_t) i * 4 + perm [k]] >> hv);
            if (unassoc && perm [k] != alane)
                v = unpremul8 (v, a8);
            dst [(size_t) i * bpp_out + k] = (uint8_t) v;
        }
    }
}

__attribute__ ((target ("avx2")))
static void
fast_pack_avx2 (const Ctx *c, const uint16_t *acc, uint32_t hv, const uint8_t perm [4], uint32_t n_px, uint8_t *dst)
{
    if (c->to->bpp == 3)
        fast_pack_body (c, acc, hv, perm, n_px, dst, 3, 0);
    else if (c->to->unassoc)
        fast_pack_body (c, acc, hv, perm, n_px, dst, 4, 1);
    else
        fast_pack_body (c, acc, hv, perm, n_px, dst, 4, 0);
}

/* ---- box filters of the fast class (round 5) ---------------------------------------------------------
 * A pixel is four 16-bit lanes in one 64-bit word, the reference's own 64bpp storage (smolscale-private.h:
 * 64bpp = 4 x 16 bits): adding a pixel to a box is one 64-bit add, an edge weight one multiply + mask
 * (lanes are <= 255 and weights <= 256, so no product leaves its lane). The planner keeps the class to spans of at
 * most 255 pixels (smol_plan.c plan_axis; smolscale.c:443-478), so lane sums stay below 2^16 and no carry crosses
 * lanes either; fast_class_ok () sends a hand-built job with a longer span down the generic road. */

/* source row -> w_in pixels of four 16-bit lanes (memory byte order; premultiplied if the source is unassociated) */
__attribute__ ((target ("avx2"), always_inline))
static inline void
fast_unpack_body (const Ctx *c, const uint8_t *src, uint64_t *out, const uint32_t bpp, const int pm)
{
    const uint32_t n = c->job->width_in;
    const uint32_t apos = c->ti->pos [3] & 3u;
    const __m256i one = _mm256_set1_epi16 (1);
    const __m128i sel24 = _mm_setr_epi8 (0, 1, 2, -128, 3, 4, 5, -128, 6, 7, 8, -128, 9, 10, 11, -128);
    const __m128i alpha24 = _mm_set1_epi32 ((int) 0xff000000u);
    uint8_t bsel [32];
    uint16_t amask [16];
    __m256i v_bsel, v_amask;
    uint32_t i, k;

    for (i = 0; i < 32; i++)
        bsel [i] = (uint8_t) ((i & 8u) + 2u * apos + (i & 1u));
    for (i = 0; i < 16; i++)
        amask [i] = (i & 3u) == apos ? 0xffffu : 0u;
    v_bsel = _mm256_loadu_si256 ((const __m256i *) bsel);
    v_amask = _mm256_loadu_si256 ((const __m256i *) amask);

    /* (24-bit rows: the 16-byte load of four pixels reaches 4 bytes past them; keep those inside the row) */
    for (i = 0; i + 4 + (bpp == 3 ? 2u : 0u) <= n; i += 4)
    {
        __m128i raw = _mm_loadu_si128 ((const __m128i *) (src + (size_t) i * bpp));
        __m256i v;

        if (bpp == 3)
            raw = _mm_or_si128 (_mm_shuffle_epi8 (raw, sel24), alpha24);
        v = _mm256_cvtepu8_epi16 (raw);
        if (pm)
        {
            const __m256i m = _mm256_add_epi16 (_mm256_shuffle_epi8 (v, v_bsel), one);
            const __m256i r = _mm256_srli_epi16 (_mm256_sub_epi16 (_mm256_mullo_epi16 (_mm256_add_epi16 (v, one), m), one), 8);
            v = _mm256_blendv_epi8 (r, v, v_amask);
        }
        _mm256_storeu_si256 ((__m256i *) (out + i), v);
    }
    for (; i < n; i++)
    {
        const uint32_t w = fast_px (src, i, bpp), a = (w >> (8 * apos)) & 0xffu;
        uint64_t v = 0;
        for (k = 0; k < 4; k++)
        {
            uint32_t ch = (w >> (8 * k)) & 0xffu;
            if (pm && k != apos)
                ch = premul8 (ch, a);
            v |= (uint64_t) ch << (16 * k);
        }
        out [i] = v;
    }
}

__attribute__ ((target ("avx2")))
static void
fast_unpack_avx2 (const Ctx *c, const uint8_t *src, uint64_t *out)
{
    if (c->ti->bpp == 3)
        fast_unpack_body (c, src, out, 3, 0);
    else if (c->ti->unassoc)
        fast_unpack_body (c, src, out, 4, 1);
    else
        fast_unpack_body (c, src, out, 4, 0);
}

/* box_scale () of n 16-bit lanes: (acc * mul + 2^23) >> 24 & 0xff, exact for any 32-bit mul (64-bit products) */
__attribute__ ((target ("avx2")))
static void
fast_box_scale_avx2 (const uint16_t *acc, uint32_t mul, uint16_t *out, size_t n)
{
    const __m256i vm = _mm256_set1_epi64x ((long long) mul), half = _mm256_set1_epi64x (1ll << 23);
    const __m256i ff = _mm256_set1_epi64x (0xff);
    size_t i = 0;

    for (; i + 16 <= n; i += 16)
    {
        const __m256i a = _mm256_loadu_si256 ((const __m256i *) (acc + i));
        __m256i w [2];
        int u;

        w [0] = _mm256_cvtepu16_epi32 (_mm256_castsi256_si128 (a));
        w [1] = _mm256_cvtepu16_epi32 (_mm256_extracti128_si256 (a, 1));
        for (u = 0; u < 2; u++)
        {
            const __m256i ev = _mm256_and_si256 (_mm256_srli_epi64 (_mm256_add_epi64 (_mm256_mul_epu32 (w [u], vm), half), 24), ff);
            const __m256i od = _mm256_and_si256 (_mm256_srli_epi64 (_mm256_add_epi64 (_mm256_mul_epu32 (_mm256_srli_epi64 (w [u], 32), vm), half), 24), ff);
            w [u] = _mm256_or_si256 (ev, _mm256_slli_epi64 (od, 32));
        }
        _mm256_storeu_si256 ((__m256i *) (out + i), _mm256_permute4x64_epi64 (_mm256_packus_epi32 (w [0], w [1]), 0xd8));
    }
    for (; i < n; i++)
        out [i] = (uint16_t) ((((uint64_t) acc [i] * mul + (1u << 23)) >> 24) & 0xffu);
}

/* horizontal box of one source row: `urow' = width_in words of scratch, `sums' = width_out words of scratch */
__attribute__ ((target ("avx2")))
static void
fast_hbox_avx2 (const Ctx *c, const uint8_t *src, uint64_t *urow, uint64_t *sums, uint16_t *out)
{
    const SmolHipJob *j = c->job;
    const uint32_t w_out = j->width_out;
    const uint64_t M = 0x00ff00ff00ff00ffull;
    uint32_t x;

    fast_unpack_avx2 (c, src, urow);
    for (x = 0; x < w_out; x++)
    {
        const uint32_t o0 = c->h_ofs [x], o1 = c->h_last [x];
        uint64_t acc = ((urow [o0] * c->h_wl [x]) >> 8) & M;
        uint32_t s;

        if (o1 > o0)
        {
            uint64_t a2 = 0;
            for (s = o0 + 1; s + 1 < o1; s += 2)
            {
                acc += urow [s];
                a2 += urow [s + 1];
            }
            if (s < o1)
                acc += urow [s];
            acc += a2 + (((urow [o1] * c->h_fr [x]) >> 8) & M);
        }
        sums [x] = acc;
    }
    fast_box_scale_avx2 ((const uint16_t *) sums, j->span_mul_x, out, (size_t) w_out * 4);
}

/* vertical box: acc (+)= (row * w) >> 8 over n 16-bit lanes (w = 256: the row itself) */
__attribute__ ((target ("avx2")))
static void
fast_vbox_add_avx2 (const uint16_t *row, uint32_t w, uint16_t *acc, int first, size_t n)
{
    const __m256i vw = _mm256_set1_epi16 ((short) w);
    size_t i = 0;

    for (; i + 16 <= n; i += 16)
    {
        __m256i v = _mm256_loadu_si256 ((const __m256i *) (row + i));
        if (w != 256u)
            v = _mm256_srli_epi16 (_mm256_mullo_epi16 (v, vw), 8);
        if (!first)
            v = _mm256_add_epi16 (v, _mm256_loadu_si256 ((const __m256i *) (acc + i)));
        _mm256_storeu_si256 ((__m256i *) (acc + i), v);
    }
    for (; i < n; i++)
    {
        const uint16_t v = (uint16_t) ((row [i] * w) >> 8);
        acc [i] = first ? v : (uint16_t) (acc [i] + v);
    }
}

/* SMOL_CPU_SIMD=0: every job down the generic 32-bit-lane road (what a CPU without AVX2 runs; the tests keep it honest) */
static int
simd_allowed (void)
{
    static int known = -1;
    int k = __atomic_load_n (&known, __ATOMIC_RELAXED);
    if (k < 0)
    {
        const char *e = getenv ("SMOL_CPU_SIMD");
        k = !(e && *e == '0') && __builtin_cpu_supports ("avx2");
        __atomic_store_n (&known, k, __ATOMIC_RELAXED);
    }
    return k;
}

static int
fast_class_ok (const Ctx *c)
{
    const SmolHipJob *j = c->job;
    /* the whole 64bpp class with a bilinear or a box filter on either axis: 3- and 4-byte pixels, either alpha convention
     * on either side (round 5; rounds 3-4: 32-bit premultiplied in and out, bilinear on both axes only) */
    if (c->wide || !simd_allowed ())
        return 0;
    if (j->filter_h == F_BOX && j->step_x > (255u << 16))
        return 0;
    if (j->filter_h == F_COPY && j->width_out != j->width_in)
        return 0;
    return !(j->filter_v == F_BOX && j->step_y > (255u << 16));
}

typedef struct
{
    const Ctx *c;
    const uint8_t *src;
    uint32_t stride;
    uint16_t *hbuf [3];                 /* cache of horizontally filtered rows (four 16-bit lanes per pixel) */
    uint32_t tag [3], next;
    uint64_t *urow, *sums;              /* box rows: the unpacked source row, the box sums                   */
}
FastRows;

/* `keep': a row handed out a moment ago that must stay valid (NULL: none) */
static const uint16_t *
fast_row_get (FastRows *fr, uint32_t r, const uint16_t *keep)
{
    uint32_t i, slot;

    for (i = 0; i < 3; i++)
        if (fr->tag [i] == r)
            return fr->hbuf [i];
    slot = fr->next;
    if (fr->hbuf [slot] == keep)
        slot = (slot + 1) % 3;
    fr->next = (slot + 1) % 3;
    if (fr->c->job->filter_h == F_BOX)
        fast_hbox_avx2 (fr->c, fr->src + (size_t) r * fr->stride, fr->urow, fr->sums, fr->hbuf [slot]);
    else if (fr->c->job->filter_h == F_BILINEAR)
        fast_hrow_avx2 (fr->c, fr->src + (size_t) r * fr->stride, fr->hbuf [slot]);
    else if (fr->c->job->filter_h == F_COPY)
        fast_unpack_avx2 (fr->c, fr->src + (size_t) r * fr->stride, (uint64_t *) (void *) fr->hbuf [slot]);    /* width_out == width_in */
    else
    {
        /* one source column: every output pixel is pixel 0 */
        uint64_t *o = (uint64_t *) (void *) fr->hbuf [slot];
        uint32_t x;
        fast_unpack_avx2 (fr->c, fr->src + (size_t) r * fr->stride, fr->urow);
        for (x = 0; x < fr->c->job->width_out; x++)
            o [x] = fr->urow [0];
    }
    fr->tag [slot] = r;
    return fr->hbuf [slot];
}

static int
fast_scale_rows (const Ctx *c, const uint8_t *src, uint32_t stride_in, uint8_t *dest, uint32_t stride_out,
                 uint32_t first_row, uint32_t n_rows)
{
    const SmolHipJob *j = c->job;
    const uint32_t w_out = j->width_out, hv = j->halvings_v, nsy = 1u << hv;
    const size_t lanes = (size_t) w_out * 4;
    const size_t box_words = j->filter_h == F_BOX ? (size_t) j->width_in + w_out : j->filter_h == F_ONE ? (size_t) j->width_in : 0;
    const int use512 = __builtin_cpu_supports ("avx512bw");
    FastRows fr;
    uint16_t *acc, *vout;
    uint32_t y, k;
    uint8_t perm [4];
    uint8_t *mem = malloc (box_words * sizeof (uint64_t) + (lanes * 5 + 64) * sizeof (uint16_t));

    if (!mem)
        return SMOL_HIP_E_INVALID;
    fr.c = c; fr.src = src; fr.stride = stride_in; fr.next = 0;
    fr.urow = (uint64_t *) mem;
    fr.sums = fr.urow + j->width_in;
    for (k = 0; k < 3; k++)
    {
        fr.hbuf [k] = (uint16_t *) (mem + box_words * sizeof (uint64_t)) + lanes * k;
        fr.tag [k] = UINT32_MAX;
    }
    acc = fr.hbuf [0] + lanes * 3;
    vout = acc + lanes;
    /* output byte k comes from the input lane that holds the same channel (a 24-bit source's alpha: the synthetic lane 3) */
    for (k = 0; k < 4; k++)
    {
        uint32_t ch;
        perm [k] = 3;
        for (ch = 0; ch < 4; ch++)
            if (c->to->pos [ch] == k)
                perm [k] = (uint8_t) ((ch == 3 && c->ti->bpp == 3) ? 3u : c->ti->pos [ch]);
    }

    for (y = first_row; y < first_row + n_rows; y++, dest += stride_out)
    {
        if (j->filter_v == F_BOX)
        {
            /* 64bpp-class rows: the last row's weight is the fraction as it is (smolscale-generic.c:2042-2069) */
            uint32_t y0, y1, w2, w1, r;

            box_span (j->step_y, j->height_in, y, &y0, &y1, &w2, &w1);
            for (r = y0; r <= y1; r++)
                fast_vbox_add_avx2 (fast_row_get (&fr, r, NULL), r == y0 ? w1 : r == y1 ? w2 : 256u, acc, r == y0, lanes);
            fast_box_scale_avx2 (acc, j->span_mul_y, vout, lanes);
            fast_pack_avx2 (c, vout, 0, perm, w_out, dest);
            continue;
        }
        if (j->filter_v != F_BILINEAR)
        {
            /* the same height (row y), or one source row (row 0) */
            fast_pack_avx2 (c, fast_row_get (&fr, j->filter_v == F_ONE ? 0u : y, NULL), 0, perm, w_out, dest);
            continue;
        }
        for (uint32_t s = 0; s < nsy; s++)
        {
            const uint16_t *top, *bot;
            uint32_t oy, F;

            bilin_tap (j->frac0_y + (uint64_t) ((y << hv) + s) * j->step_y, j->height_in, &oy, &F);
            top = fast_row_get (&fr, oy, NULL);
            bot = fast_row_get (&fr, oy + 1, top);
            if (use512)
                fast_vblend_avx512 (top, bot, F, acc, s == 0, lanes);
            else
                fast_vblend_avx2 (top, bot, F, acc, s == 0, lanes);
        }
        fast_pack_avx2 (c, acc, hv, perm, w_out, dest);
    }
    free (mem);
    return 0;
}

/* ---- the 128bpp class with SIMD (round 5) ---------------------------------------------------------------
 * Four 32-bit lanes per pixel in the generic road's own order (R, G, B, alpha lane; its unpack_row () and pack_row () serve
 * the last pixels of a row): the filters -- bilinear with halvings, box -- run on one pixel per 128-bit register, the
 * vertical pass, the unpackers and the packers on two, table look-ups as gathers (the gamma tables from 32-bit copies),
 * the 64-bit products of the inverse-alpha multiplies as even / odd 32 x 32 -> 64 multiplies. */

/* horizontal bilinear: (((p - q) * F) >> 8) + q, masked, summed over the samples, halved */
__attribute__ ((target ("avx2")))
static void
wide_hrow_bilinear_avx2 (const Ctx *c, const uint32_t *u, uint32_t *out)
{
    const SmolHipJob *j = c->job;
    const uint32_t w_out = j->width_out, h = j->halvings_h, ns = 1u << h;
    const uint32_t *ofs = c->h_ofs;
    const uint16_t *f = c->h_f;
    const __m128i mask = _mm_set1_epi32 ((int) c->mask);
    const __m128i shift = _mm_cvtsi32_si128 ((int) h);
    uint32_t x, s;

    for (x = 0; x < w_out; x++, out += 4)
    {
        __m128i acc = _mm_setzero_si128 ();
        for (s = 0; s < ns; s++, ofs++, f++)
        {
            const __m128i p = _mm_loadu_si128 ((const __m128i *) (u + (size_t) *ofs * 4));
            const __m128i q = _mm_loadu_si128 ((const __m128i *) (u + (size_t) *ofs * 4 + 4));
            const __m128i d = _mm_mullo_epi32 (_mm_sub_epi32 (p, q), _mm_set1_epi32 ((int) *f));
            acc = _mm_add_epi32 (acc, _mm_and_si128 (_mm_add_epi32 (_mm_srai_epi32 (d, 8), q), mask));
        }
        _mm_storeu_si128 ((__m128i *) out, _mm_and_si128 (_mm_srl_epi32 (acc, shift), mask));
    }
}

/* box_scale () of four 32-bit lanes: (acc * mul + 2^23) >> 24, masked to 16 bits (64-bit products) */
__attribute__ ((target ("avx2")))
static inline __m128i
wide_box_scale4 (__m128i acc, __m256i vm)
{
    const __m256i half = _mm256_set1_epi64x (1ll << 23);
    const __m256i w = _mm256_cvtepu32_epi64 (acc);
    const __m256i r = _mm256_srli_epi64 (_mm256_add_epi64 (_mm256_mul_epu32 (w, vm), half), 24);
    /* the low 32 bits of each 64-bit lane -> four 32-bit lanes */
    const __m256i packed = _mm256_permutevar8x32_epi32 (r, _mm256_setr_epi32 (0, 2, 4, 6, 0, 2, 4, 6));
    return _mm_and_si128 (_mm256_castsi256_si128 (packed), _mm_set1_epi32 (0xffff));
}

__attribute__ ((target ("avx2")))
static void
wide_hrow_box_avx2 (const Ctx *c, const uint32_t *u, uint32_t *out)
{
    const SmolHipJob *j = c->job;
    const uint32_t w_out = j->width_out;
    const __m128i mask = _mm_set1_epi32 ((int) c->mask);
    const __m256i vm = _mm256_set1_epi64x ((long long) j->span_mul_x);
    uint32_t x, s;

    for (x = 0; x < w_out; x++, out += 4)
    {
        const uint32_t o0 = c->h_ofs [x], o1 = c->h_last [x];
        __m128i acc = _mm_and_si128 (_mm_srli_epi32 (_mm_mullo_epi32 (_mm_loadu_si128 ((const __m128i *) (u + (size_t) o0 * 4)),
                                                                      _mm_set1_epi32 (c->h_wl [x])), 8), mask);
        if (o1 > o0)
        {
            __m128i a2 = _mm_setzero_si128 ();
            for (s = o0 + 1; s + 1 < o1; s += 2)
            {
                acc = _mm_add_epi32 (acc, _mm_loadu_si128 ((const __m128i *) (u + (size_t) s * 4)));
                a2 = _mm_add_epi32 (a2, _mm_loadu_si128 ((const __m128i *) (u + (size_t) s * 4 + 4)));
            }
            if (s < o1)
                acc = _mm_add_epi32 (acc, _mm_loadu_si128 ((const __m128i *) (u + (size_t) s * 4)));
            acc = _mm_add_epi32 (_mm_add_epi32 (acc, a2),
                                 _mm_and_si128 (_mm_srli_epi32 (_mm_mullo_epi32 (_mm_loadu_si128 ((const __m128i *) (u + (size_t) o1 * 4)),
                                                                                 _mm_set1_epi32 (c->h_fr [x])), 8), mask));
        }
        _mm_storeu_si128 ((__m128i *) out, wide_box_scale4 (acc, vm));
    }
}

/* vertical bilinear over n 32-bit lanes: acc (+)= lerp (top, bot, F) */
__attribute__ ((target ("avx2")))
static void
wide_vblend_avx2 (const uint32_t *top, const uint32_t *bot, uint32_t F, uint32_t lane_mask, uint32_t *acc, int first, size_t n)
{
    const __m256i vf = _mm256_set1_epi32 ((int) F), mask = _mm256_set1_epi32 ((int) lane_mask);
    size_t i = 0;

    for (; i + 8 <= n; i += 8)
    {
        const __m256i t = _mm256_loadu_si256 ((const __m256i *) (top + i));
        const __m256i b = _mm256_loadu_si256 ((const __m256i *) (bot + i));
        __m256i v = _mm256_and_si256 (_mm256_add_epi32 (_mm256_srai_epi32 (_mm256_mullo_epi32 (_mm256_sub_epi32 (t, b), vf), 8), b), mask);
        if (!first)
            v = _mm256_add_epi32 (v, _mm256_loadu_si256 ((const __m256i *) (acc + i)));
        _mm256_storeu_si256 ((__m256i *) (acc + i), v);
    }
    for (; i < n; i++)
    {
        const uint32_t v = lerp (top [i], bot [i], F, lane_mask);
        acc [i] = first ? v : acc [i] + v;
    }
}

/* vertical box over n 32-bit lanes: acc (+)= (row * w) >> 8, masked (w = 256: the row itself) */
__attribute__ ((target ("avx2")))
static void
wide_vbox_add_avx2 (const uint32_t *row, uint32_t w, uint32_t lane_mask, uint32_t *acc, int first, size_t n)
{
    const __m256i vw = _mm256_set1_epi32 ((int) w), mask = _mm256_set1_epi32 ((int) lane_mask);
    size_t i = 0;

    for (; i + 8 <= n; i += 8)
    {
        __m256i v = _mm256_loadu_si256 ((const __m256i *) (row + i));
        if (w != 256u)
            v = _mm256_and_si256 (_mm256_srli_epi32 (_mm256_mullo_epi32 (v, vw), 8), mask);
        if (!first)
            v = _mm256_add_epi32 (v, _mm256_loadu_si256 ((const __m256i *) (acc + i)));
        _mm256_storeu_si256 ((__m256i *) (acc + i), v);
    }
    for (; i < n; i++)
    {
        const uint32_t v = w != 256u ? ((row [i] * w) >> 8) & lane_mask : row [i];
        acc [i] = first ? v : acc [i] + v;
    }
}

/* 32-bit copies of the two gamma tables: a gather reads four bytes per index */
static uint32_t wide_from_srgb32 [256], wide_to_srgb32 [2048];

__attribute__ ((constructor))
static void
wide_tables_init (void)
{
    uint32_t i;
    for (i = 0; i < 256; i++)
        wide_from_srgb32 [i] = smol_from_srgb_lut [i];
    for (i = 0; i < 2048; i++)
        wide_to_srgb32 [i] = smol_to_srgb_lut [i];
}

/* (x * y) >> sh per 32-bit lane with 64-bit products; the low 32 bits of each result */
__attribute__ ((target ("avx2"), always_inline))
static inline __m256i
wide_mul_shr (__m256i x, __m256i y, const int sh)
{
    const __m256i ev = _mm256_srli_epi64 (_mm256_mul_epu32 (x, y), sh);
    const __m256i od = _mm256_srli_epi64 (_mm256_mul_epu32 (_mm256_srli_epi64 (x, 32), _mm256_srli_epi64 (y, 32)), sh);
    return _mm256_blend_epi32 (ev, _mm256_slli_epi64 (od, 32), 0xaa);
}

/* unpack_row () two pixels at a time, one specialisation per (bytes per pixel, alpha convention, storage form):
 * the same formulas, lanes in the same order (R, G, B, alpha lane) */
__attribute__ ((target ("avx2"), always_inline))
static inline void
wide_unpack_body (const Ctx *c, const uint8_t *src, uint32_t n, uint32_t *out,
                  const uint32_t bpp, const int unassoc, const int p16, const int linear)
{
    const TypeInfo *ti = c->ti;
    const uint8_t pa = bpp == 4 ? ti->pos [3] : 0x80;
    const __m128i sel = _mm_setr_epi8 ((char) ti->pos [0], (char) ti->pos [1], (char) ti->pos [2], (char) pa,
                                       (char) (bpp + ti->pos [0]), (char) (bpp + ti->pos [1]), (char) (bpp + ti->pos [2]),
                                       (char) (bpp == 4 ? 4 + ti->pos [3] : 0x80),
                                       -128, -128, -128, -128, -128, -128, -128, -128);
    const __m128i opaque = bpp == 3 ? _mm_setr_epi8 (0, 0, 0, -1, 0, 0, 0, -1, 0, 0, 0, 0, 0, 0, 0, 0) : _mm_setzero_si128 ();
    const __m256i alpha_lanes = _mm256_setr_epi32 (0, 0, 0, -1, 0, 0, 0, -1);
    const __m256i ff = _mm256_set1_epi32 (0xff), one = _mm256_set1_epi32 (1);
    uint32_t i;

    /* (the 8-byte load of two 24-bit pixels reaches 2 bytes past them; keep those inside the row) */
    for (i = 0; i + 2 + (bpp == 3 ? 1u : 0u) <= n; i += 2)
    {
        const __m128i raw = _mm_or_si128 (_mm_shuffle_epi8 (_mm_loadl_epi64 ((const __m128i *) (src + (size_t) i * bpp)), sel), opaque);
        const __m256i v = _mm256_cvtepu8_epi32 (raw);
        const __m256i a = _mm256_shuffle_epi32 (v, 0xff);
        __m256i col, al;

        if (p16)
        {
            col = linear ? _mm256_i32gather_epi32 ((const int *) wide_from_srgb32, v, 4) : v;
            col = _mm256_mullo_epi32 (col, a);
            al = _mm256_or_si256 (_mm256_slli_epi32 (a, 8), _mm256_set1_epi32 (0x80));
        }
        else if (linear)
        {
            __m256i x = v, l;
            if (bpp == 4 && !unassoc)
                x = _mm256_and_si256 (wide_mul_shr (v, _mm256_i32gather_epi32 ((const int *) smol_inv_p8_lut, a, 4), 13), ff);
            else if (bpp == 3)
                x = _mm256_and_si256 (wide_mul_shr (v, _mm256_set1_epi32 ((int) smol_inv_p8_lut [255]), 13), ff);
            l = _mm256_i32gather_epi32 ((const int *) wide_from_srgb32, x, 4);
            /* premul8l (): ((l + 1) * ((a << 3) + 1) - 1) >> 11, 11 bits */
            col = _mm256_and_si256 (_mm256_srli_epi32 (_mm256_sub_epi32 (_mm256_mullo_epi32 (_mm256_add_epi32 (l, one),
                                                                                              _mm256_add_epi32 (_mm256_slli_epi32 (a, 3), one)), one), 11),
                                    _mm256_set1_epi32 (0x7ff));
            al = a;
        }
        else
        {
            col = unassoc ? _mm256_and_si256 (_mm256_srli_epi32 (_mm256_sub_epi32 (_mm256_mullo_epi32 (_mm256_add_epi32 (v, one), _mm256_add_epi32 (a, one)), one), 8), ff)
                          : v;
            al = a;
        }
        _mm256_storeu_si256 ((__m256i *) (out + (size_t) i * 4), _mm256_blendv_epi8 (col, al, alpha_lanes));
    }
    if (i < n)
        unpack_row (c, src + (size_t) i * bpp, n - i, out + (size_t) i * 4);
}

__attribute__ ((target ("avx2")))
static void
wide_unpack_avx2 (const Ctx *c, const uint8_t *src, uint32_t n, uint32_t *out)
{
    const int u = c->ti->unassoc;
    if (c->ti->bpp == 3)
    {
        if (c->p16)          { if (c->linear) wide_unpack_body (c, src, n, out, 3, 0, 1, 1); else wide_unpack_body (c, src, n, out, 3, 0, 1, 0); }
        else if (c->linear)  wide_unpack_body (c, src, n, out, 3, 0, 0, 1);
        else                 wide_unpack_body (c, src, n, out, 3, 0, 0, 0);
    }
    else if (c->p16)         { if (c->linear) wide_unpack_body (c, src, n, out, 4, 1, 1, 1); else wide_unpack_body (c, src, n, out, 4, 1, 1, 0); }
    else if (c->linear)      { if (u) wide_unpack_body (c, src, n, out, 4, 1, 0, 1); else wide_unpack_body (c, src, n, out, 4, 0, 0, 1); }
    else                     { if (u) wide_unpack_body (c, src, n, out, 4, 1, 0, 0); else wide_unpack_body (c, src, n, out, 4, 0, 0, 0); }
}

/* pack_row () two pixels at a time, one specialisation per (bytes per pixel, alpha convention, storage form) */
__attribute__ ((target ("avx2"), always_inline))
static inline void
wide_pack_body (const Ctx *c, const uint32_t *in, uint32_t n, uint8_t *dst,
                const uint32_t bpp, const int unassoc, const int p16, const int linear)
{
    const TypeInfo *to = c->to;
    const int quirk24 = c->quirk24;
    const __m256i ff = _mm256_set1_epi32 (0xff), m11 = _mm256_set1_epi32 (0x7ff), one = _mm256_set1_epi32 (1);
    uint8_t sel8 [16];
    uint32_t i, k;
    __m128i sel;

    /* bytes R G B A | R G B A of two pixels -> the output type's order, 24-bit pixels back to back */
    memset (sel8, 0x80, sizeof sel8);
    for (k = 0; k < 2; k++)
    {
        uint32_t ch;
        for (ch = 0; ch < bpp; ch++)
            sel8 [k * bpp + to->pos [ch]] = (uint8_t) (4 * k + ch);
    }
    sel = _mm_loadu_si128 ((const __m128i *) sel8);

    /* (the 8-byte store of two 24-bit pixels reaches 2 bytes past them; keep those inside the row) */
    for (i = 0; i + 2 + (bpp == 3 ? 1u : 0u) <= n; i += 2)
    {
        const __m256i v = _mm256_loadu_si256 ((const __m256i *) (in + (size_t) i * 4));
        const __m256i araw = _mm256_shuffle_epi32 (v, 0xff);
        const __m256i a = _mm256_and_si256 (p16 ? _mm256_srli_epi32 (araw, 8) : araw, ff);
        __m256i o;
        __m128i b;

        if (p16)
        {
            if (linear)
            {
                const __m256i t = _mm256_and_si256 (wide_mul_shr (v, _mm256_i32gather_epi32 ((const int *) smol_inv_p16l_lut, a, 4), 19), m11);
                o = _mm256_i32gather_epi32 ((const int *) wide_to_srgb32, t, 4);
            }
            else
                o = _mm256_and_si256 (wide_mul_shr (v, _mm256_i32gather_epi32 ((const int *) smol_inv_p16_lut, a, 4), 16), ff);
        }
        else if (linear)
        {
            const __m256i u = _mm256_and_si256 (wide_mul_shr (v, _mm256_i32gather_epi32 ((const int *) smol_inv_p8l_lut, a, 4), 10), m11);
            if (bpp == 3)
                o = _mm256_i32gather_epi32 ((const int *) wide_to_srgb32, quirk24 ? _mm256_and_si256 (v, m11) : u, 4);
            else
            {
                o = _mm256_i32gather_epi32 ((const int *) wide_to_srgb32, u, 4);
                if (!unassoc)
                    o = _mm256_and_si256 (_mm256_srli_epi32 (_mm256_sub_epi32 (_mm256_mullo_epi32 (_mm256_add_epi32 (o, one), _mm256_add_epi32 (a, one)), one), 8), ff);
            }
        }
        else
            o = unassoc ? _mm256_and_si256 (wide_mul_shr (v, _mm256_i32gather_epi32 ((const int *) smol_inv_p8_lut, a, 4), 13), ff)
                        : _mm256_and_si256 (v, ff);
        o = _mm256_blend_epi32 (o, a, 0x88);            /* the alpha lanes */
        o = _mm256_packus_epi32 (o, o);                 /* per half: four 16-bit values, twice */
        o = _mm256_packus_epi16 (o, o);                 /* per half: four bytes, four times */
        b = _mm_shuffle_epi8 (_mm_unpacklo_epi32 (_mm256_castsi256_si128 (o), _mm256_extracti128_si256 (o, 1)), sel);
        _mm_storel_epi64 ((__m128i *) (dst + (size_t) i * bpp), b);
    }
    if (i < n)
        pack_row (c, in + (size_t) i * 4, n - i, dst + (size_t) i * bpp);
}

__attribute__ ((target ("avx2")))
static void
wide_pack_avx2 (const Ctx *c, const uint32_t *in, uint32_t n, uint8_t *dst)
{
    const int u = c->to->unassoc;
    if (c->to->bpp == 3)
    {
        if (c->p16)          { if (c->linear) wide_pack_body (c, in, n, dst, 3, 0, 1, 1); else wide_pack_body (c, in, n, dst, 3, 0, 1, 0); }
        else if (c->linear)  wide_pack_body (c, in, n, dst, 3, 0, 0, 1);
        else                 wide_pack_body (c, in, n, dst, 3, 0, 0, 0);
    }
    else if (c->p16)         { if (c->linear) wide_pack_body (c, in, n, dst, 4, 1, 1, 1); else wide_pack_body (c, in, n, dst, 4, 1, 1, 0); }
    else if (c->linear)      { if (u) wide_pack_body (c, in, n, dst, 4, 1, 0, 1); else wide_pack_body (c, in, n, dst, 4, 0, 0, 1); }
    else                     { if (u) wide_pack_body (c, in, n, dst, 4, 1, 0, 0); else wide_pack_body (c, in, n, dst, 4, 0, 0, 0); }
}

static int
wide_class_ok (const Ctx *c)
{
    const SmolHipJob *j = c->job;
    if (!c->wide || !simd_allowed ())
        return 0;
    return !(j->filter_h == F_COPY && j->width_out != j->width_in);
}

typedef struct
{
    const Ctx *c;
    const uint8_t *src;
    uint32_t stride;
    uint32_t *urow, *hbuf [3];
    uint32_t tag [3], next;
}
WideRows;

static const uint32_t *
wide_row_get (WideRows *wr, uint32_t r, const uint32_t *keep)
{
    const Ctx *c = wr->c;
    uint32_t i, slot;

    for (i = 0; i < 3; i++)
        if (wr->tag [i] == r)
            return wr->hbuf [i];
    slot = wr->next;
    if (wr->hbuf [slot] == keep)
        slot = (slot + 1) % 3;
    wr->next = (slot + 1) % 3;
    wide_unpack_avx2 (c, wr->src + (size_t) r * wr->stride, c->job->width_in, wr->urow);
    if (c->job->filter_h == F_BOX)
        wide_hrow_box_avx2 (c, wr->urow, wr->hbuf [slot]);
    else if (c->job->filter_h == F_BILINEAR)
        wide_hrow_bilinear_avx2 (c, wr->urow, wr->hbuf [slot]);
    else if (c->job->filter_h == F_COPY)
        memcpy (wr->hbuf [slot], wr->urow, (size_t) c->job->width_out * 16);
    else
    {
        uint32_t x;
        for (x = 0; x < c->job->width_out; x++)
            memcpy (wr->hbuf [slot] + (size_t) x * 4, wr->urow, 16);
    }
    wr->tag [slot] = r;
    return wr->hbuf [slot];
}

__attribute__ ((target ("avx2")))
static int
wide_scale_rows (const Ctx *c, const uint8_t *src, uint32_t stride_in, uint8_t *dest, uint32_t stride_out,
                 uint32_t first_row, uint32_t n_rows)
{
    const SmolHipJob *j = c->job;
    const uint32_t w_out = j->width_out, hv = j->halvings_v, nsy = 1u << hv;
    const size_t lanes = (size_t) w_out * 4, lanes_in = (size_t) j->width_in * 4;
    const __m256i vmy = _mm256_set1_epi64x ((long long) j->span_mul_y);
    WideRows wr;
    uint32_t *mem, *acc, *vout;
    uint32_t y, k;
    size_t i;

    mem = malloc ((lanes_in + lanes * 5 + 32) * sizeof (uint32_t));
    if (!mem)
        return SMOL_HIP_E_INVALID;
    wr.c = c; wr.src = src; wr.stride = stride_in; wr.next = 0;
    wr.urow = mem;
    for (k = 0; k < 3; k++)
    {
        wr.hbuf [k] = mem + lanes_in + 8 + lanes * k;
        wr.tag [k] = UINT32_MAX;
    }
    acc = mem + lanes_in + 8 + lanes * 3;
    vout = acc + lanes;

    for (y = first_row; y < first_row + n_rows; y++, dest += stride_out)
    {
        if (j->filter_v == F_BOX)
        {
            /* 128bpp-class rows weight the last row by w2 - 1 and skip it when w2 == 0 (smolscale-generic.c:2236-2253) */
            uint32_t y0, y1, w2, w1, r, last_w, y_end;

            box_span (j->step_y, j->height_in, y, &y0, &y1, &w2, &w1);
            last_w = w2 ? w2 - 1 : 0;
            y_end = w2 == 0 ? y1 - 1 : y1;
            if (y_end < y0 || y_end == UINT32_MAX)
                memset (acc, 0, lanes * sizeof (uint32_t));
            for (r = y0; r <= y_end && y_end != UINT32_MAX; r++)
                wide_vbox_add_avx2 (wide_row_get (&wr, r, NULL), r == y0 ? w1 : r == y1 ? last_w : 256u, c->mask, acc, r == y0, lanes);
            for (i = 0; i < lanes; i += 4)
                _mm_storeu_si128 ((__m128i *) (vout + i), wide_box_scale4 (_mm_loadu_si128 ((const __m128i *) (acc + i)), vmy));
        }
        else if (j->filter_v != F_BILINEAR)
        {
            /* the same height (row y), or one source row (row 0) */
            wide_pack_avx2 (c, wide_row_get (&wr, j->filter_v == F_ONE ? 0u : y, NULL), w_out, dest);
            continue;
        }
        else
        {
            const __m256i mask = _mm256_set1_epi32 ((int) c->mask);
            const __m128i shift = _mm_cvtsi32_si128 ((int) hv);
            uint32_t s;

            for (s = 0; s < nsy; s++)
            {
                const uint32_t *top, *bot;
                uint32_t oy, F;

                bilin_tap (j->frac0_y + (uint64_t) ((y << hv) + s) * j->step_y, j->height_in, &oy, &F);
                top = wide_row_get (&wr, oy, NULL);
                bot = wide_row_get (&wr, oy + 1, top);
                wide_vblend_avx2 (top, bot, F, c->mask, acc, s == 0, lanes);
            }
            for (i = 0; i + 8 <= lanes; i += 8)
                _mm256_storeu_si256 ((__m256i *) (vout + i),
                                     _mm256_and_si256 (_mm256_srl_epi32 (_mm256_loadu_si256 ((const __m256i *) (acc + i)), shift), mask));
            for (; i < lanes; i++)
                vout [i] = (acc [i] >> hv) & c->mask;
        }
        wide_pack_avx2 (c, vout, w_out, dest);
    }
    free (mem);
    return 0;
}

#endif /* SMOL_CPU_X86 */

/* ---- tables ---------------------------------------------------------------------------------------- */

static int
build_tables (Ctx *c)
{
    const SmolHipJob *j = c->job;
    uint32_t x;

    if (j->filter_h == F_BILINEAR)
    {
        const uint32_t n = j->width_out << j->halvings_h;
        c->h_ofs = malloc ((size_t) n * sizeof (uint32_t));
        c->h_f = malloc ((size_t) n * sizeof (uint16_t));
        if (!c->h_ofs || !c->h_f)
            return SMOL_HIP_E_INVALID;
        for (x = 0; x < n; x++)
        {
            uint32_t o, F;
            bilin_tap (j->frac0_x + (uint64_t) x * j->step_x, j->width_in, &o, &F);
            c->h_ofs [x] = o;
            c->h_f [x] = (uint16_t) F;
        }
    }
    else if (j->filter_h == F_BOX)
    {
        const uint32_t n = j->width_out;
        c->h_ofs = malloc ((size_t) n * sizeof (uint32_t));
        c->h_last = malloc ((size_t) n * sizeof (uint32_t));
        c->h_wl = malloc ((size_t) n * sizeof (uint16_t));
        c->h_fr = malloc ((size_t) n * sizeof (uint16_t));
        if (!c->h_ofs || !c->h_last || !c->h_wl || !c->h_fr)
            return SMOL_HIP_E_INVALID;
        for (x = 0; x < n; x++)
        {
            uint32_t o0, o1, fr, wl;
            box_span (j->step_x, j->width_in, x, &o0, &o1, &fr, &wl);
            c->h_ofs [x] = o0; c->h_last [x] = o1; c->h_wl [x] = (uint16_t) wl; c->h_fr [x] = (uint16_t) fr;
        }
    }
    return 0;
}

static void
free_tables (Ctx *c)
{
    free (c->h_ofs); free (c->h_f); free (c->h_last); free (c->h_wl); free (c->h_fr);
}

/* ---- entry points ------------------------------------------------------------------------------------ */

/* A hand-built job (the ABI is reachable through ctypes) must not index outside the tables or the image */
static int
job_valid (const SmolHipJob *j)
{
    if (j->pixel_type_in >= 10 || j->pixel_type_out >= 10)
        return 0;
    if (!j->width_in || !j->height_in || !j->width_out || !j->height_out)
        return 0;
    if (j->filter_h > F_BOX || j->filter_v > F_BOX || j->halvings_h > 2 || j->halvings_v > 2)
        return 0;
    if ((j->filter_h == F_BILINEAR && j->width_in < 2) || (j->filter_v == F_BILINEAR && j->height_in < 2))
        return 0;                       /* bilin_tap clamps to dim_in - 2 */
    if ((j->filter_h == F_COPY && j->width_in != j->width_out) || (j->filter_v == F_COPY && j->height_in != j->height_out))
        return 0;
    if ((j->filter_h == F_BOX && (j->step_x >> 16) > j->width_in) || (j->filter_v == F_BOX && (j->step_y >> 16) > j->height_in))
        return 0;
    if ((j->filter_h == F_BOX && j->step_x < 0x10000u) || (j->filter_v == F_BOX && j->step_y < 0x10000u))
        return 0;
    /* the last box must end inside the image: box k ends at (k + 1) * step (16.16). The planner's step,
     * floor (dim_in * 65536 / dim_out), always does; 64 -> 64 with step 2.0 would read up to pixel 126 */
    if ((j->filter_h == F_BOX && (((uint64_t) j->width_out * j->step_x) >> 16) > j->width_in)
        || (j->filter_v == F_BOX && (((uint64_t) j->height_out * j->step_y) >> 16) > j->height_in))
        return 0;
    return 1;
}

struct SmolCpuPlan
{
    SmolHipJob job;
    Ctx c;
};

int
smol_cpu_plan_create (const SmolHipJob *job, SmolCpuPlan **plan_out)
{
    SmolCpuPlan *p;
    int rc;

    if (!job || !plan_out || !job_valid (job))
        return SMOL_HIP_E_INVALID;
    p = calloc (1, sizeof (*p));
    if (!p)
        return SMOL_HIP_E_INVALID;
    p->job = *job;
    p->c.job = &p->job;
    p->c.ti = &k_types [job->pixel_type_in];
    p->c.to = &k_types [job->pixel_type_out];
    p->c.wide = job->wide; p->c.p16 = job->premul16; p->c.linear = job->linear; p->c.quirk24 = job->quirk24;
    p->c.mask = p->c.wide ? 0x00ffffffu : 0xffu;
    rc = build_tables (&p->c);
    if (rc)
    {
        free_tables (&p->c);
        free (p);
        return rc;
    }
    *plan_out = p;
    return 0;
}

void
smol_cpu_plan_destroy (SmolCpuPlan *plan)
{
    if (!plan)
        return;
    free_tables (&plan->c);
    free (plan);
}

int
smol_cpu_scale_rows (const SmolHipJob *job,
                     const void *pixels_in, uint32_t rowstride_in,
                     void *dest, uint32_t rowstride_out,
                     uint32_t first_row, uint32_t n_rows)
{
    SmolCpuPlan *plan;
    int rc;

    if (!job || !pixels_in || !dest)
        return SMOL_HIP_E_INVALID;
    if (n_rows == 0)
        return job_valid (job) ? 0 : SMOL_HIP_E_INVALID;
    rc = smol_cpu_plan_create (job, &plan);
    if (rc)
        return rc;
    rc = smol_cpu_plan_scale_rows (plan, pixels_in, rowstride_in, dest, rowstride_out, first_row, n_rows);
    smol_cpu_plan_destroy (plan);
    return rc;
}

/* one strip of output rows on the calling thread (the plan is immutable: any number of strips may run at once) */
SMOL_CPU_CLONES
static int
strip_scale_rows (const SmolCpuPlan *plan,
                  const void *pixels_in, uint32_t rowstride_in,
                  void *dest, uint32_t rowstride_out,
                  uint32_t first_row, uint32_t n_rows)
{
    const Ctx *cp = &plan->c;
    const SmolHipJob *job = &plan->job;
    RowSource rs;
    uint32_t *mem, *vout, *acc;
    uint32_t y, k, x;
    size_t row_lanes_in, row_lanes_out;
    uint8_t *drow = dest;

#ifdef SMOL_CPU_X86
    if (fast_class_ok (cp))
        return fast_scale_rows (cp, pixels_in, rowstride_in, dest, rowstride_out, first_row, n_rows);
    if (wide_class_ok (cp))
        return wide_scale_rows (cp, pixels_in, rowstride_in, dest, rowstride_out, first_row, n_rows);
#endif

    row_lanes_in = (size_t) job->width_in * 4;
    row_lanes_out = (size_t) job->width_out * 4;
    mem = malloc ((row_lanes_in + row_lanes_out * 5) * sizeof (uint32_t));
    if (!mem)
        return SMOL_HIP_E_INVALID;
    rs.c = cp; rs.src = pixels_in; rs.stride = rowstride_in; rs.next = 0;
    rs.urow = mem;
    for (k = 0; k < 3; k++)
    {
        rs.hrow [k] = mem + row_lanes_in + row_lanes_out * k;
        rs.tag [k] = UINT32_MAX;
    }
    vout = mem + row_lanes_in + row_lanes_out * 3;
    acc = vout + row_lanes_out;

    for (y = first_row; y < first_row + n_rows; y++, drow += rowstride_out)
    {
        const uint32_t mask = cp->mask;

        if (job->filter_v == F_BILINEAR)
        {
            const uint32_t hv = job->halvings_v, nsy = 1u << hv;
            uint32_t s;

            for (s = 0; s < nsy; s++)
            {
                uint32_t oy, F;
                const uint32_t *top, *bot;

                bilin_tap (job->frac0_y + (uint64_t) ((y << hv) + s) * job->step_y, job->height_in, &oy, &F);
                top = hrow_get (&rs, oy, NULL);
                bot = hrow_get (&rs, oy + 1, top);
                for (x = 0; x < row_lanes_out; x++)
                {
                    const uint32_t v = lerp (top [x], bot [x], F, mask);
                    acc [x] = s ? acc [x] + v : v;
                }
            }
            for (x = 0; x < row_lanes_out; x++)
                vout [x] = (acc [x] >> hv) & mask;
        }
        else if (job->filter_v == F_BOX)
        {
            uint32_t y0, y1, w2, w1, r;
            uint32_t last_w, y_end;

            box_span (job->step_y, job->height_in, y, &y0, &y1, &w2, &w1);
            /* 128bpp-class rows weight the last row by w2 - 1 and skip it when w2 == 0
             * (smolscale-generic.c:2236-2253); 64bpp-class rows use w2 as is (:2042-2069) */
            last_w = cp->wide ? (w2 ? w2 - 1 : 0) : w2;
            y_end = (cp->wide && w2 == 0) ? y1 - 1 : y1;
            memset (acc, 0, row_lanes_out * sizeof (uint32_t));
            for (r = y0; r <= y_end; r++)
            {
                const uint32_t *h = hrow_get (&rs, r, NULL);
                if (r == y0)
                    for (x = 0; x < row_lanes_out; x++) acc [x] += ((h [x] * w1) >> 8) & mask;
                else if (r == y1)
                    for (x = 0; x < row_lanes_out; x++) acc [x] += ((h [x] * last_w) >> 8) & mask;
                else
                    for (x = 0; x < row_lanes_out; x++) acc [x] += h [x];
            }
            for (x = 0; x < row_lanes_out; x++)
                vout [x] = box_scale (cp, acc [x], job->span_mul_y);
        }
        else
        {
            const uint32_t *h = hrow_get (&rs, job->filter_v == F_ONE ? 0u : y, NULL);
            memcpy (vout, h, row_lanes_out * sizeof (uint32_t));
        }
        pack_row (cp, vout, job->width_out, drow);
    }

    free (mem);
    return 0;
}

/* ---- a call's rows over the host cores ------------------------------------------------------------------
 * Output rows are independent given the source (the reference's own harness cuts an image into one strip per
 * worker: test.c:838-883): a call of enough rows is cut into strips that run on a small persistent pool, the
 * calling thread taking its share. SMOL_CPU_THREADS = threads per call (default: the online cores, at most 24;
 * 1 = never leave the calling thread). Each strip re-filters the one or two source rows it shares with its
 * neighbour -- a bilinear halo -- and nothing else is shared. */
#include <pthread.h>
#include <unistd.h>

typedef struct
{
    const SmolCpuPlan *plan;
    const uint8_t *src;
    uint8_t *dest;
    uint32_t stride_in, stride_out, first, n, n_strips;
    uint32_t next;                      /* next strip to claim (under pool_mtx) */
    uint32_t done;                      /* strips finished */
    int rc;
} CpuCall;

static pthread_mutex_t pool_mtx = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t pool_work = PTHREAD_COND_INITIALIZER, pool_done = PTHREAD_COND_INITIALIZER;
static CpuCall *pool_call;              /* the call being served (one at a time; others run on their own thread) */
static uint32_t pool_threads;           /* helpers alive */

static void
call_run_strip (CpuCall *c, uint32_t i)
{
    const uint32_t per = (c->n + c->n_strips - 1) / c->n_strips;
    const uint32_t a = i * per, b = a + per < c->n ? a + per : c->n;
    if (a < b)
    {
        const int rc = strip_scale_rows (c->plan, c->src, c->stride_in, c->dest + (size_t) a * c->stride_out, c->stride_out,
                                         c->first + a, b - a);
        if (rc)
        {
            pthread_mutex_lock (&pool_mtx);
            c->rc = rc;
            pthread_mutex_unlock (&pool_mtx);
        }
    }
}

static void *
pool_main (void *arg)
{
    (void) arg;
    pthread_mutex_lock (&pool_mtx);
    for (;;)
    {
        CpuCall *c = pool_call;
        if (c && c->next < c->n_strips)
        {
            const uint32_t i = c->next++;
            pthread_mutex_unlock (&pool_mtx);
            call_run_strip (c, i);
            pthread_mutex_lock (&pool_mtx);
            if (++c->done == c->n_strips)
                pthread_cond_broadcast (&pool_done);
            continue;
        }
        pthread_cond_wait (&pool_work, &pool_mtx);
    }
    return NULL;
}

static uint32_t
cpu_threads_wanted (void)
{
    static uint32_t cached;
    if (!cached)
    {
        const char *e = getenv ("SMOL_CPU_THREADS");
        long v = e && *e ? strtol (e, NULL, 10) : 0;
        if (v <= 0)
        {
            /* (on the GPU box's 256 hardware threads: 24 threads 0.48 ms per 4096^2 -> 1024^2 frame, 32: 0.54, 64: 1.31,
             * 128: 1.76 -- waking helpers costs more than they carry; profiles/r05_cpu_backend_classes.txt) */
            v = sysconf (_SC_NPROCESSORS_ONLN);
            if (v > 24) v = 24;
        }
        if (v < 1) v = 1;
        if (v > 256) v = 256;
        cached = (uint32_t) v;
    }
    return cached;
}

int
smol_cpu_plan_scale_rows (const SmolCpuPlan *plan,
                          const void *pixels_in, uint32_t rowstride_in,
                          void *dest, uint32_t rowstride_out,
                          uint32_t first_row, uint32_t n_rows)
{
    const SmolHipJob *job;
    uint32_t want;
    uint64_t work;

    if (!plan || !pixels_in || !dest)
        return SMOL_HIP_E_INVALID;
    job = &plan->job;
    if (n_rows == 0)
        return 0;
    if (first_row >= job->height_out || n_rows > job->height_out - first_row)
        return SMOL_HIP_E_INVALID;

    /* worth cutting up? (pixels read + written by the call, roughly) */
    work = (uint64_t) n_rows * job->width_out + (uint64_t) n_rows * job->width_in * (job->height_in / (job->height_out ? job->height_out : 1) + 1);
    want = cpu_threads_wanted ();
    if (want > n_rows / 4)
        want = n_rows / 4;
    /* at least 128 K pixels of work per thread (640 x 480 -> 320 x 240: 3 threads 0.04 ms, 32 threads 0.12) */
    if ((uint64_t) want > work / 131072u)
        want = (uint32_t) (work / 131072u);
    if (want < 2 || work < 262144)
        return strip_scale_rows (plan, pixels_in, rowstride_in, dest, rowstride_out, first_row, n_rows);

    {
        CpuCall call;
        uint32_t i;

        call.plan = plan; call.src = pixels_in; call.dest = dest;
        call.stride_in = rowstride_in; call.stride_out = rowstride_out;
        call.first = first_row; call.n = n_rows;
        call.n_strips = want * 2;       /* twice the threads: the last strips even out the finish */
        if (call.n_strips > n_rows)
            call.n_strips = n_rows;
        call.next = 0; call.done = 0; call.rc = 0;

        pthread_mutex_lock (&pool_mtx);
        if (pool_call)
        {
            /* the pool serves another caller's call: this one runs where it is (callers that thread themselves --
             * the reference's harness does -- must not queue behind each other) */
            pthread_mutex_unlock (&pool_mtx);
            return strip_scale_rows (plan, pixels_in, rowstride_in, dest, rowstride_out, first_row, n_rows);
        }
        while (pool_threads < want - 1)
        {
            pthread_t t;
            pthread_attr_t at;
            pthread_attr_init (&at);
            pthread_attr_setdetachstate (&at, PTHREAD_CREATE_DETACHED);
            if (pthread_create (&t, &at, pool_main, NULL))
            {
                pthread_attr_destroy (&at);
                break;                  /* no thread to be had: the caller works through the strips itself */
            }
            pthread_attr_destroy (&at);
            pool_threads++;
        }
        pool_call = &call;
        pthread_cond_broadcast (&pool_work);
        /* the caller takes strips too */
        while (call.next < call.n_strips)
        {
            i = call.next++;
            pthread_mutex_unlock (&pool_mtx);
            call_run_strip (&call, i);
            pthread_mutex_lock (&pool_mtx);
            call.done++;
        }
        while (call.done < call.n_strips)
            pthread_cond_wait (&pool_done, &pool_mtx);
        pool_call = NULL;
        pthread_mutex_unlock (&pool_mtx);
        return call.rc;
    }
}
