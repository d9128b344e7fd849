/* smol_oracle.c -- TEST INFRASTRUCTURE ONLY (see smol_oracle.h).
 *
 * A CPU restatement of the reference resampler's algorithm, written per colour
 * channel with ordinary integer arithmetic instead of the reference's 64-bit
 * SWAR lanes. Every function cites the reference file:line it follows
 * (paths relative to /root/reference). The restatement is pinned against the
 * compiled reference (oracle/_ref) by tests/test_oracle_vs_reference.py.
 *
 * Conventions: a pixel in flight is four uint32 channels in canonical order
 * R, G, B, A. ">>" on a signed value is a floor shift (arithmetic). */

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "smol_oracle.h"
#include "smol_oracle_luts.h"

typedef struct { uint32_t c [4]; } OPix;   /* R, G, B, A */

/* ------------------------------------------------------------------ *
 * Pixel types: byte position of each canonical channel in memory.    *
 * smolscale.h:16-36 (enum order), smolscale.c:45-58 (storage/alpha). *
 * ------------------------------------------------------------------ */

typedef struct
{
    uint8_t bpp;            /* bytes per pixel */
    int8_t pos [4];         /* byte index of R, G, B, A; -1 = absent */
    uint8_t unassoc;        /* unassociated alpha */
}
OTypeMeta;

static const OTypeMeta type_meta [10] =
{
    { 4, { 0, 1, 2, 3 }, 0 },   /* RGBA8 premultiplied */
    { 4, { 2, 1, 0, 3 }, 0 },   /* BGRA8 premultiplied */
    { 4, { 1, 2, 3, 0 }, 0 },   /* ARGB8 premultiplied */
    { 4, { 3, 2, 1, 0 }, 0 },   /* ABGR8 premultiplied */
    { 4, { 0, 1, 2, 3 }, 1 },   /* RGBA8 unassociated  */
    { 4, { 2, 1, 0, 3 }, 1 },   /* BGRA8 unassociated  */
    { 4, { 1, 2, 3, 0 }, 1 },   /* ARGB8 unassociated  */
    { 4, { 3, 2, 1, 0 }, 1 },   /* ABGR8 unassociated  */
    { 3, { 0, 1, 2, -1 }, 0 },  /* RGB8 (counts as PREMUL8, smolscale.c:56) */
    { 3, { 2, 1, 0, -1 }, 0 }   /* BGR8 (smolscale.c:57) */
};

/* ------------------------------------------------------------------ *
 * Planner                                                            *
 * ------------------------------------------------------------------ */

/* smolscale.c:427-478 pick_filter_params() */
static void
plan_axis (uint32_t dim_in, uint32_t dim_out, uint32_t with_srgb,
           uint32_t *filter, uint32_t *storage, uint32_t *halvings, uint32_t *dim_bilin)
{
    *dim_bilin = dim_out;
    *storage = with_srgb ? 3 : 2;
    *halvings = 0;

    if (dim_in > dim_out * 255)
    {
        *filter = ORACLE_FILTER_BOX;
        *storage = 3;
    }
    else if (dim_in > dim_out * 8)
    {
        *filter = ORACLE_FILTER_BOX;
    }
    else if (dim_in == 1)
    {
        *filter = ORACLE_FILTER_ONE;
    }
    else if (dim_in == dim_out)
    {
        *filter = ORACLE_FILTER_COPY;
    }
    else
    {
        uint32_t n = 0, d = dim_out;

        for (;;)
        {
            d *= 2;
            if (d >= dim_in)
                break;
            n++;
        }

        *dim_bilin = dim_out << n;
        *filter = ORACLE_FILTER_BILINEAR_0H + n;
        *halvings = n;
    }
}

/* smolscale-generic.c:82-91: the box normaliser computed at the top of
 * precalc_boxes_array(). */
static uint32_t
box_span_mul (uint32_t dim_in, uint32_t dim_out)
{
    uint64_t step = ((uint64_t) dim_in * 65536) / (uint64_t) dim_out;
    uint64_t stride = step / 65536;
    uint64_t f = (step / 256) % 256;
    uint64_t a = ((uint64_t) 65536 * 256) * 255;
    uint64_t b = (stride * 255) + ((f * 255) / 256);

    return (uint32_t) ((a + (b / 2)) / b);
}

/* smolscale.c:816-870 smol_scale_init() + :725-783 get_implementations().
 * quirk24: which of the two 128bpp->24bpp packers find_repacks() (:647-719)
 * lands on. With 128bpp storage the unpackers leave channels in mid order
 * {3,2,1,4} for memory types RGBA/ARGB/BGR8 and {1,2,3,4} for BGRA/ABGR/RGB8
 * (little-endian remap :63-75 + do_reorder :620-645); the `123' packer is first
 * in the table (smolscale-generic.c:2369-2380) and matches BGR8 from the former
 * and RGB8 from the latter. Only its LINEAR/PREMUL8 variant (:922-935) differs
 * in behaviour from the `321' one (:1010-1023). */
int
smol_oracle_plan (uint32_t type_in, uint32_t w_in, uint32_t h_in,
                  uint32_t type_out, uint32_t w_out, uint32_t h_out,
                  uint32_t with_srgb, OraclePlan *p)
{
    uint32_t st_h, st_v;

    if (type_in >= 10 || type_out >= 10 || !w_in || !h_in || !w_out || !h_out
        || w_in > 65535 || h_in > 65535 || w_out > 65535 || h_out > 65535)
        return -1;

    memset (p, 0, sizeof (*p));
    plan_axis (w_in, w_out, with_srgb, &p->filter_h, &st_h, &p->width_halvings, &p->width_bilin_out);
    plan_axis (h_in, h_out, with_srgb, &p->filter_v, &st_v, &p->height_halvings, &p->height_bilin_out);
    p->storage_type = st_h > st_v ? st_h : st_v;                         /* :862 */
    if (p->filter_h == ORACLE_FILTER_BOX)
        p->span_mul_x = box_span_mul (w_in, w_out);
    if (p->filter_v == ORACLE_FILTER_BOX)
        p->span_mul_y = box_span_mul (h_in, h_out);
    p->gamma_type = with_srgb ? 1 : 0;                                   /* :846 */

    if (type_meta [type_in].unassoc && type_meta [type_out].unassoc)     /* :751-758 */
    {
        p->premul16 = 1;
        p->storage_type = 3;
    }

    if (w_in > w_out * 8191 || h_in > h_out * 8191)                      /* :760-770 */
        p->gamma_type = 0;

    if (type_meta [type_out].bpp == 3)
    {
        int mid_is_3214 = (type_in == 0 || type_in == 4 || type_in == 2 || type_in == 6 || type_in == 9);
        p->quirk24 = mid_is_3214 ? (type_out == 9) : (type_out == 8);
    }

    return 0;
}

/* ------------------------------------------------------------------ *
 * Precalculation (the reference's own array layout)                  *
 * ------------------------------------------------------------------ */

#define BILIN_MUL ((uint64_t) 65536 * 65536)     /* smolscale-private.h:70 */
#define BOXES_MUL ((uint64_t) 65536 * 256)       /* smolscale-private.h:69 */

/* smolscale-generic.c:14-66 precalc_bilinear_array() */
static void
precalc_bilinear (uint16_t *array, uint32_t dim_in, uint32_t dim_out, int absolute)
{
    uint64_t frac, step;
    uint16_t *pu16 = array;
    uint16_t last_ofs = 0;

    if (dim_in > dim_out)
    {
        step = ((uint64_t) dim_in * BILIN_MUL) / dim_out;
        frac = (step - BILIN_MUL) / 2;
    }
    else
    {
        step = ((uint64_t) (dim_in - 1) * BILIN_MUL) / (dim_out > 1 ? (dim_out - 1) : 1);
        frac = 0;
    }

    do
    {
        uint16_t ofs = (uint16_t) (frac / BILIN_MUL);

        if (ofs >= dim_in - 1)
            break;

        *(pu16++) = absolute ? ofs : (uint16_t) (ofs - last_ofs);
        *(pu16++) = (uint16_t) (256 - ((frac / (BILIN_MUL / 256)) % 256));
        frac += step;
        last_ofs = ofs;
    }
    while (--dim_out);

    while (dim_out)
    {
        *(pu16++) = absolute ? (uint16_t) (dim_in - 2) : (uint16_t) ((dim_in - 2) - last_ofs);
        *(pu16++) = 0;
        dim_out--;
        last_ofs = (uint16_t) (dim_in - 2);
    }
}

/* smolscale-generic.c:68-135 precalc_boxes_array() */
static void
precalc_boxes (uint16_t *array, uint32_t *span_mul, uint32_t dim_in, uint32_t dim_out, int absolute)
{
    uint64_t frac = 0, step;
    uint16_t *pu16 = array;
    uint16_t ofs = 0, next_ofs;
    uint64_t f, stride, a, b;

    step = ((uint64_t) dim_in * 65536) / (uint64_t) dim_out;
    stride = step / 65536;
    f = (step / 256) % 256;

    a = BOXES_MUL * 255;
    b = (stride * 255) + ((f * 255) / 256);
    *span_mul = (uint32_t) ((a + (b / 2)) / b);

    do
    {
        frac += step;
        next_ofs = (uint16_t) (frac / 65536);

        if (ofs >= dim_in - 1)
        {
            ofs = (uint16_t) (dim_in - 1);
            break;
        }

        if (next_ofs > dim_in - 1)
        {
            next_ofs = (uint16_t) (dim_in - 1);
            if (next_ofs <= ofs)
                break;
        }

        stride = (uint64_t) next_ofs - ofs - 1;
        f = (frac / 256) % 256;

        *(pu16++) = absolute ? ofs : (uint16_t) stride;
        *(pu16++) = (uint16_t) f;

        ofs = next_ofs;
    }
    while (--dim_out);

    while (dim_out)
    {
        *(pu16++) = absolute ? ofs : 0;
        *(pu16++) = 0;
        dim_out--;
    }

    *(pu16++) = absolute ? ofs : 0;
    *(pu16++) = 0;
}

/* ------------------------------------------------------------------ *
 * Job context                                                        *
 * ------------------------------------------------------------------ */

typedef struct
{
    OraclePlan plan;
    const uint8_t *pixels_in;
    uint32_t type_in, type_out;
    uint32_t w_in, h_in, stride_in;
    uint32_t w_out, h_out, stride_out;
    int is128, linear, p16;
    uint32_t lane_mask;            /* 0xff (64bpp) or 0x00ffffff (128bpp) */
    uint32_t box_mask;             /* 0xff (64bpp) or 0xffff (128bpp) */
    uint16_t *precalc_x, *precalc_y;
    /* scratch */
    OPix *unpacked;                /* w_in */
    OPix *hrow [2];                /* w_out each: cached horizontally scaled rows */
    uint32_t hrow_y [2];
    OPix *tmp;                     /* w_out */
    uint64_t *acc;                 /* w_out * 4 */
}
OCtx;

/* smolscale-generic.c:137-179 init_horizontal()/init_vertical() */
static int
octx_init (OCtx *o, const void *pixels_in, uint32_t type_in,
           uint32_t w_in, uint32_t h_in, uint32_t stride_in,
           uint32_t type_out, uint32_t w_out, uint32_t h_out, uint32_t stride_out,
           uint32_t with_srgb)
{
    memset (o, 0, sizeof (*o));
    if (smol_oracle_plan (type_in, w_in, h_in, type_out, w_out, h_out, with_srgb, &o->plan))
        return -1;

    o->pixels_in = pixels_in;
    o->type_in = type_in; o->type_out = type_out;
    o->w_in = w_in; o->h_in = h_in; o->stride_in = stride_in;
    o->w_out = w_out; o->h_out = h_out; o->stride_out = stride_out;
    o->is128 = (o->plan.storage_type == 3);
    o->linear = (o->plan.gamma_type == 1);
    o->p16 = (int) o->plan.premul16;
    o->lane_mask = o->is128 ? 0x00ffffffu : 0xffu;
    o->box_mask = o->is128 ? 0xffffu : 0xffu;

    o->precalc_x = calloc ((size_t) (o->plan.width_bilin_out + 2) * 2 + (size_t) (w_out + 2) * 2, sizeof (uint16_t));
    o->precalc_y = calloc ((size_t) (o->plan.height_bilin_out + 2) * 2 + (size_t) (h_out + 2) * 2, sizeof (uint16_t));

    if (o->plan.filter_h == ORACLE_FILTER_BOX)
        precalc_boxes (o->precalc_x, &o->plan.span_mul_x, w_in, w_out, 0);
    else if (o->plan.filter_h >= ORACLE_FILTER_BILINEAR_0H)
        precalc_bilinear (o->precalc_x, w_in, o->plan.width_bilin_out, 0);

    if (o->plan.filter_v == ORACLE_FILTER_BOX)
        precalc_boxes (o->precalc_y, &o->plan.span_mul_y, h_in, h_out, 1);
    else if (o->plan.filter_v >= ORACLE_FILTER_BILINEAR_0H)
        precalc_bilinear (o->precalc_y, h_in, o->plan.height_bilin_out, 1);

    o->unpacked = malloc (sizeof (OPix) * ((size_t) w_in + 2));
    o->hrow [0] = malloc (sizeof (OPix) * (size_t) w_out);
    o->hrow [1] = malloc (sizeof (OPix) * (size_t) w_out);
    o->tmp = malloc (sizeof (OPix) * (size_t) w_out);
    o->acc = malloc (sizeof (uint64_t) * 4 * (size_t) w_out);
    o->hrow_y [0] = o->hrow_y [1] = 0xffffffffu;
    return 0;
}

static void
octx_free (OCtx *o)
{
    free (o->precalc_x); free (o->precalc_y);
    free (o->unpacked); free (o->hrow [0]); free (o->hrow [1]); free (o->tmp); free (o->acc);
}

/* ------------------------------------------------------------------ *
 * Unpack: memory bytes -> internal premultiplied channels            *
 * ------------------------------------------------------------------ */

/* smolscale-generic.c:238-244 premul_u_to_p8_64bpp(), per 16-bit lane */
static inline uint32_t
premul8 (uint32_t c, uint32_t a)
{
    return (((c + 1) * (a + 1) - 1) >> 8) & 0xff;
}

/* smolscale-generic.c:261-269 premul_ul_to_p8l_128bpp(), per 32-bit lane */
static inline uint32_t
premul8l (uint32_t l, uint32_t a)
{
    return (((l + 1) * ((a << 3) + 1) - 1) >> 11) & 0x7ff;
}

/* smolscale-generic.c:227-236 / :246-259 unpremul_p8_to_u_*() */
static inline uint32_t
unpremul8 (uint32_t c, uint32_t a)
{
    return (uint32_t) ((((uint64_t) c * oracle_inv_p8_lut [a]) >> 13) & 0xff);
}

/* One row. Cases, in the order of the reference's unpackers:
 *  64bpp  premul / 24bpp : copy                     smolscale-generic.c:353-413
 *  64bpp  unassoc        : premul8                  :415-464
 *  128bpp p16            : c*a, A=(a<<8)|0x80       :616-625, 708-717 (+from_srgb :627-660, 719-752)
 *  128bpp linear, premul : unpremul8 -> from_srgb -> premul8l   :481-495, 513-535, 554-568
 *  128bpp linear, unassoc: from_srgb -> premul8l    :588-614, 680-706
 *  128bpp compressed     : as 64bpp                 :470-479, 497-511, 537-552, 570-586, 662-678 */
static void
unpack_row (const OCtx *o, const uint8_t *row, OPix *out)
{
    const OTypeMeta *m = &type_meta [o->type_in];
    uint32_t x;

    for (x = 0; x < o->w_in; x++, row += m->bpp)
    {
        uint32_t a = m->pos [3] >= 0 ? row [m->pos [3]] : 0xff;
        int i;

        for (i = 0; i < 3; i++)
        {
            uint32_t c = row [m->pos [i]];

            if (o->is128 && o->p16)
            {
                if (o->linear)
                    c = oracle_from_srgb_lut [c];
                c = c * a;
            }
            else if (o->is128 && o->linear)
            {
                if (!m->unassoc)
                    c = unpremul8 (c, a);
                c = premul8l (oracle_from_srgb_lut [c], a);
            }
            else if (m->unassoc)
            {
                c = premul8 (c, a);
            }

            out [x].c [i] = c;
        }

        out [x].c [3] = (o->is128 && o->p16) ? ((a << 8) | 0x80) : a;
    }
}

/* ------------------------------------------------------------------ *
 * Pack: internal channels -> memory bytes                            *
 * ------------------------------------------------------------------ */

/* Cases:
 *  64bpp  -> premul/24bpp: bytes                    smolscale-generic.c:758-770, 789-797, 814-822, 839-847, 864-870, 883-891
 *  64bpp  -> unassoc     : unpremul8                :872-881, 892-901
 *  128bpp p16            : a=(A>>8)&0xff; inv_p16 / inv_p16l + to_srgb   :290-318, 1135-1158
 *  128bpp compressed     : & 0xff, or unpremul8     :912-920, 937-949, 1000-1008, 1025-1037, 1087-1095, 1111-1121
 *  128bpp linear -> 32bpp premul : unpremul8l -> to_srgb -> premul8      :1096-1109
 *  128bpp linear -> 32bpp unassoc: unpremul8l -> to_srgb                 :1122-1134
 *  128bpp linear -> 24bpp: `321' packer unpremul8l -> to_srgb (:1010-1023);
 *                          `123' packer passes the premultiplied row to to_srgb (:922-935, quirk24). */
static void
pack_row (const OCtx *o, const OPix *in, uint8_t *row)
{
    const OTypeMeta *m = &type_meta [o->type_out];
    uint32_t x;

    for (x = 0; x < o->w_out; x++, row += m->bpp)
    {
        uint32_t a;
        int i;

        if (o->is128 && o->p16)
            a = (in [x].c [3] >> 8) & 0xff;
        else
            a = in [x].c [3] & 0xff;

        for (i = 0; i < 3; i++)
        {
            uint32_t v = in [x].c [i];
            uint32_t c;

            if (!o->is128)
            {
                c = m->unassoc ? unpremul8 (v, a) : (v & 0xff);
            }
            else if (o->p16)
            {
                if (o->linear)
                    c = oracle_to_srgb_lut [(((uint64_t) v * oracle_inv_p16l_lut [a]) >> 19) & 0x7ff];
                else
                    c = (uint32_t) ((((uint64_t) v * oracle_inv_p16_lut [a]) >> 16) & 0xff);
            }
            else if (!o->linear)
            {
                c = m->unassoc ? unpremul8 (v, a) : (v & 0xff);
            }
            else
            {
                uint32_t u = (uint32_t) ((((uint64_t) v * oracle_inv_p8l_lut [a]) >> 10) & 0x7ff);

                if (m->bpp == 3)
                {
                    /* The reference indexes its 2048-entry table with the raw lane
                     * here; lanes are <= 0x7ff for every value unpack_row can make. */
                    c = o->plan.quirk24 ? oracle_to_srgb_lut [v & 0x7ff] : oracle_to_srgb_lut [u];
                }
                else if (m->unassoc)
                {
                    c = oracle_to_srgb_lut [u];
                }
                else
                {
                    c = premul8 (oracle_to_srgb_lut [u], a);
                }
            }

            row [m->pos [i]] = (uint8_t) c;
        }

        if (m->pos [3] >= 0)
            row [m->pos [3]] = (uint8_t) a;
    }
}

/* ------------------------------------------------------------------ *
 * Horizontal filters                                                 *
 * ------------------------------------------------------------------ */

/* One bilinear tap: smolscale-generic.c:1317 / :1704  ((p-q)*F >> 8) + q, masked. */
static inline uint32_t
lerp (uint32_t p, uint32_t q, uint32_t F, uint32_t mask)
{
    int64_t d = ((int64_t) p - (int64_t) q) * (int64_t) F;
    return (uint32_t) ((d >> 8) + (int64_t) q) & mask;
}

/* smolscale-generic.c:1231-1261 scale_64bpp() / scale_128bpp_half(), one lane.
 * (64bpp lanes are 16 bits wide in the reference: `accum & 0x0000ffff0000ffff'.) */
static inline uint32_t
box_scale (const OCtx *o, uint64_t acc, uint32_t mul)
{
    if (!o->is128)
        acc &= 0xffff;
    else
        acc &= 0xffffffffu;
    return (uint32_t) (((acc * mul + (BOXES_MUL / 2)) / BOXES_MUL) & o->box_mask);
}

/* smolscale-generic.c:1290-1425 interp_horizontal_bilinear_*() */
static void
hfilter_bilinear (const OCtx *o, const OPix *in, OPix *out)
{
    const uint16_t *pc = o->precalc_x;
    uint32_t h = o->plan.width_halvings;
    uint32_t pos = 0, k, j;
    int i;

    for (k = 0; k < o->w_out; k++)
    {
        uint32_t acc [4] = { 0, 0, 0, 0 };

        for (j = 0; j < (1u << h); j++)
        {
            uint32_t F;

            pos += *(pc++);
            F = *(pc++);
            for (i = 0; i < 4; i++)
                acc [i] += lerp (in [pos].c [i], in [pos + 1].c [i], F, o->lane_mask);
        }

        for (i = 0; i < 4; i++)
            out [k].c [i] = (acc [i] >> h) & o->lane_mask;
    }
}

/* smolscale-generic.c:1427-1556 interp_horizontal_boxes_*() */
static void
hfilter_box (const OCtx *o, const OPix *in, OPix *out)
{
    const uint16_t *pc = o->precalc_x;
    uint32_t mul = o->plan.span_mul_x;
    uint32_t pp = 0, n, F, k, t;
    uint64_t acc [4] = { 0, 0, 0, 0 };
    uint32_t carry [4];
    int i;

    for (i = 0; i < 4; i++)                               /* weight_pixel (p, 256) */
        carry [i] = (uint32_t) ((((uint64_t) in [pp].c [i] * 256) >> 8) & o->lane_mask);
    pp++;
    n = *(pc++);

    for (k = 0; k + 1 < o->w_out; k++)
    {
        for (t = 0; t < n; t++, pp++)
            for (i = 0; i < 4; i++)
                acc [i] += in [pp].c [i];

        F = *(pc++);
        n = *(pc++);

        for (i = 0; i < 4; i++)
        {
            uint64_t r = in [pp].c [i];
            uint64_t s = r * F;
            uint32_t q = (uint32_t) ((s >> 8) & o->lane_mask);

            acc [i] += carry [i] + q;
            carry [i] = (uint32_t) ((((r << 8) - r - s) >> 8) & o->lane_mask);
            out [k].c [i] = box_scale (o, acc [i], mul);
            acc [i] = 0;
        }
        pp++;
    }

    for (t = 0; t < n; t++, pp++)
        for (i = 0; i < 4; i++)
            acc [i] += in [pp].c [i];

    F = *pc;
    for (i = 0; i < 4; i++)
    {
        uint32_t q = 0;

        if (F > 0)
            q = (uint32_t) ((((uint64_t) in [pp].c [i] * F) >> 8) & o->lane_mask);
        acc [i] += carry [i] + q;
        out [o->w_out - 1].c [i] = box_scale (o, acc [i], mul);
    }
}

/* smolscale-generic.c:1613-1642 scale_horizontal(): unpack one input row and run
 * the horizontal filter (:1558-1611 for ONE/COPY). */
static void
scale_horizontal (OCtx *o, uint32_t y_in, OPix *out)
{
    const uint8_t *row = o->pixels_in + (size_t) o->stride_in * y_in;
    uint32_t k;

    unpack_row (o, row, o->unpacked);

    switch (o->plan.filter_h)
    {
        case ORACLE_FILTER_COPY:
            memcpy (out, o->unpacked, sizeof (OPix) * o->w_out);
            break;
        case ORACLE_FILTER_ONE:
            for (k = 0; k < o->w_out; k++)
                out [k] = o->unpacked [0];
            break;
        case ORACLE_FILTER_BOX:
            hfilter_box (o, o->unpacked, out);
            break;
        default:
            hfilter_bilinear (o, o->unpacked, out);
            break;
    }
}

/* ------------------------------------------------------------------ *
 * Vertical filters, one output row each                              *
 * ------------------------------------------------------------------ */

/* Row cache of smolscale-generic.c:1648-1682 update_vertical_ctx_bilinear():
 * slot 0 holds row `y', slot 1 row `y+1'. */
static void
fetch_row_pair (OCtx *o, uint32_t y)
{
    if (o->hrow_y [0] == y && o->hrow_y [1] == y + 1)
        return;

    if (o->hrow_y [1] == y && o->hrow_y [0] + 1 == y)
    {
        OPix *t = o->hrow [0];
        o->hrow [0] = o->hrow [1];
        o->hrow [1] = t;
        scale_horizontal (o, y + 1, o->hrow [1]);
    }
    else
    {
        scale_horizontal (o, y, o->hrow [0]);
        scale_horizontal (o, y + 1, o->hrow [1]);
    }

    o->hrow_y [0] = y;
    o->hrow_y [1] = y + 1;
}

/* smolscale-generic.c:1841-2007 scale_outrow_bilinear_*(): store / add / final */
static void
vfilter_bilinear (OCtx *o, uint32_t y_out, OPix *out)
{
    uint32_t h = o->plan.height_halvings;
    uint32_t j, k;
    int i;

    for (k = 0; k < o->w_out; k++)
        for (i = 0; i < 4; i++)
            o->acc [k * 4 + i] = 0;

    for (j = 0; j < (1u << h); j++)
    {
        uint32_t bi = (y_out << h) + j;
        uint32_t y = o->precalc_y [bi * 2];
        uint32_t F = o->precalc_y [bi * 2 + 1];

        fetch_row_pair (o, y);
        for (k = 0; k < o->w_out; k++)
            for (i = 0; i < 4; i++)
                o->acc [k * 4 + i] += lerp (o->hrow [0][k].c [i], o->hrow [1][k].c [i], F, o->lane_mask);
    }

    for (k = 0; k < o->w_out; k++)
        for (i = 0; i < 4; i++)
            out [k].c [i] = (uint32_t) (o->acc [k * 4 + i] >> h) & o->lane_mask;
}

/* smolscale-generic.c:2026-2161 (64bpp) and :2163-2260 (128bpp) scale_outrow_box_*().
 * Stateless form: the in-place carry row of :2064-2065 equals
 * weight_edge_row (row, 255 - F) of :2026-2040. */
static void
vfilter_box (OCtx *o, uint32_t y_out, OPix *out)
{
    const uint16_t *py = o->precalc_y;
    uint32_t y0 = py [y_out * 2];
    uint32_t y1 = py [(y_out + 1) * 2];
    uint32_t w1 = (y_out == 0) ? 256 : 255 - py [y_out * 2 - 1];
    uint32_t w2 = py [y_out * 2 + 1];
    uint32_t mul = o->plan.span_mul_y;
    uint32_t y, k;
    int i;

    scale_horizontal (o, y0, o->tmp);
    for (k = 0; k < o->w_out; k++)
        for (i = 0; i < 4; i++)
            o->acc [k * 4 + i] = (((uint64_t) o->tmp [k].c [i] * w1) >> 8) & o->lane_mask;

    for (y = y0 + 1; y < y1; y++)
    {
        scale_horizontal (o, y, o->tmp);
        for (k = 0; k < o->w_out; k++)
            for (i = 0; i < 4; i++)
                o->acc [k * 4 + i] += o->tmp [k].c [i];
    }

    if (!o->is128)
    {
        /* :2095-2107 -- the last row is skipped (zeros) only when w2 == 0 and it
         * would be out of bounds. */
        if (w2 || y1 < o->h_in)
        {
            scale_horizontal (o, y1, o->tmp);
            for (k = 0; k < o->w_out; k++)
                for (i = 0; i < 4; i++)
                    o->acc [k * 4 + i] += (((uint64_t) o->tmp [k].c [i] * w2) >> 8) & o->lane_mask;
        }
    }
    else if (w2 > 0)
    {
        /* :2236-2253 -- row index is where the whole-row loop stopped; weight w-1 */
        uint32_t ylast = (y1 > y0 + 1) ? y1 : y0 + 1;

        scale_horizontal (o, ylast, o->tmp);
        for (k = 0; k < o->w_out; k++)
            for (i = 0; i < 4; i++)
                o->acc [k * 4 + i] += (((uint64_t) o->tmp [k].c [i] * (w2 - 1)) >> 8) & o->lane_mask;
    }

    for (k = 0; k < o->w_out; k++)
        for (i = 0; i < 4; i++)
            out [k].c [i] = box_scale (o, o->acc [k * 4 + i], mul);
}

/* smolscale.c:492-546 scale_outrow()/do_rows() + smolscale-generic.c:2262-2318 */
int
smol_oracle_scale_rows (const void *pixels_in, uint32_t type_in,
                        uint32_t w_in, uint32_t h_in, uint32_t rowstride_in,
                        void *dest, uint32_t type_out,
                        uint32_t w_out, uint32_t h_out, uint32_t rowstride_out,
                        uint32_t with_srgb,
                        uint32_t first_row, uint32_t n_rows)
{
    OCtx o;
    OPix *vout;
    uint32_t y;

    if (octx_init (&o, pixels_in, type_in, w_in, h_in, rowstride_in,
                   type_out, w_out, h_out, rowstride_out, with_srgb))
        return -1;
    if (first_row + n_rows > h_out)
    {
        octx_free (&o);
        return -1;
    }

    vout = malloc (sizeof (OPix) * (size_t) w_out);

    for (y = first_row; y < first_row + n_rows; y++)
    {
        uint8_t *row_out = (uint8_t *) dest + (size_t) rowstride_out * (y - first_row);

        switch (o.plan.filter_v)
        {
            case ORACLE_FILTER_COPY:
                scale_horizontal (&o, y, vout);
                break;
            case ORACLE_FILTER_ONE:
                scale_horizontal (&o, 0, vout);
                break;
            case ORACLE_FILTER_BOX:
                vfilter_box (&o, y, vout);
                break;
            default:
                vfilter_bilinear (&o, y, vout);
                break;
        }

        pack_row (&o, vout, row_out);
    }

    free (vout);
    octx_free (&o);
    return 0;
}

uint32_t
smol_oracle_precalc (uint32_t w_in, uint32_t h_in, uint32_t w_out, uint32_t h_out,
                     uint32_t axis, uint16_t *dest, uint32_t max_u16)
{
    OCtx o;
    uint32_t n = 0, i;
    const uint16_t *src;
    static uint8_t dummy [16];

    if (octx_init (&o, dummy, 0, w_in, h_in, w_in * 4, 0, w_out, h_out, w_out * 4, 0))
        return 0;

    if (axis == 0)
    {
        src = o.precalc_x;
        if (o.plan.filter_h == ORACLE_FILTER_BOX)
            n = (w_out + 1) * 2;
        else if (o.plan.filter_h >= ORACLE_FILTER_BILINEAR_0H)
            n = o.plan.width_bilin_out * 2;
    }
    else
    {
        src = o.precalc_y;
        if (o.plan.filter_v == ORACLE_FILTER_BOX)
            n = (h_out + 1) * 2;
        else if (o.plan.filter_v >= ORACLE_FILTER_BILINEAR_0H)
            n = o.plan.height_bilin_out * 2;
    }

    if (n > max_u16)
        n = max_u16;
    for (i = 0; i < n; i++)
        dest [i] = src [i];

    octx_free (&o);
    return n;
}

/* ------------------------------------------------------------------ *
 * Synthetic data + digest helpers (shared by tests and bench)        *
 * ------------------------------------------------------------------ */

void
smol_oracle_fill (void *buf, uint64_t n_bytes, uint64_t seed)
{
    uint8_t *p = buf;
    uint64_t x = seed ? seed : 0x9E3779B97F4A7C15ULL;
    uint64_t i;

    for (i = 0; i < n_bytes; i += 8)
    {
        uint64_t j, n = n_bytes - i < 8 ? n_bytes - i : 8;

        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        for (j = 0; j < n; j++)
            p [i + j] = (uint8_t) (x >> (8 * j));
    }
}

void
smol_oracle_make_valid_premul (void *pixels, uint32_t type, uint32_t width,
                               uint32_t height, uint32_t rowstride)
{
    const OTypeMeta *m;
    uint32_t x, y;
    int i;

    if (type >= 10)
        return;
    m = &type_meta [type];
    if (m->unassoc || m->bpp != 4)
        return;

    for (y = 0; y < height; y++)
    {
        uint8_t *row = (uint8_t *) pixels + (size_t) rowstride * y;

        for (x = 0; x < width; x++, row += 4)
        {
            uint32_t a = row [m->pos [3]];

            for (i = 0; i < 3; i++)
                row [m->pos [i]] = (uint8_t) (row [m->pos [i]] % (a + 1));
        }
    }
}

uint64_t
smol_oracle_fnv1a (const void *pixels, uint32_t row_bytes, uint32_t height, uint32_t rowstride)
{
    uint64_t h = 0xcbf29ce484222325ULL;
    uint32_t x, y;

    for (y = 0; y < height; y++)
    {
        const uint8_t *row = (const uint8_t *) pixels + (size_t) rowstride * y;

        for (x = 0; x < row_bytes; x++)
        {
            h ^= row [x];
            h *= 0x100000001b3ULL;
        }
    }

    return h;
}
