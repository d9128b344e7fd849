/* ref_probe.c -- TEST INFRASTRUCTURE (oracle/): our own probe, linked into the
 * oracle/_ref build of the real reference. It creates a reference context via the
 * public smol_scale_new() (smolscale.h:56-60) and reports the planner decisions
 * the reference stored in its private context (smolscale-private.h:280-312), so
 * tests can pin this repo's planner against the reference's.
 *
 * Compiled with -I/root/reference; nothing from the reference is copied here. */

#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include "smolscale-private.h"

typedef struct
{
    uint32_t filter_h, filter_v;        /* SmolFilterType, smolscale-private.h:100-116 */
    uint32_t storage_type;              /* SmolStorageType, :90-98 */
    uint32_t gamma_type;                /* SmolGammaType, :160-166 */
    uint32_t width_bilin_out, height_bilin_out;
    uint32_t width_halvings, height_halvings;
    uint32_t span_mul_x, span_mul_y;
}
RefProbePlan;

void
ref_probe_plan (uint32_t type_in, uint32_t w_in, uint32_t h_in,
                uint32_t type_out, uint32_t w_out, uint32_t h_out,
                uint32_t with_srgb, RefProbePlan *plan_out)
{
    static char dummy_in [16], dummy_out [16];
    SmolScaleCtx *ctx;

    ctx = smol_scale_new (dummy_in, (SmolPixelType) type_in, w_in, h_in, w_in * 4,
                          dummy_out, (SmolPixelType) type_out, w_out, h_out, w_out * 4,
                          (uint8_t) with_srgb);

    memset (plan_out, 0, sizeof (*plan_out));
    plan_out->filter_h = ctx->filter_h;
    plan_out->filter_v = ctx->filter_v;
    plan_out->storage_type = ctx->storage_type;
    plan_out->gamma_type = ctx->gamma_type;
    plan_out->width_bilin_out = ctx->width_bilin_out;
    plan_out->height_bilin_out = ctx->height_bilin_out;
    /* halvings / span_mul are only meaningful for the matching filter class
     * (pick_filter_params writes halvings only on the bilinear branch,
     * smolscale.c:457-477; calloc'd ctx so they read 0 otherwise). */
    plan_out->width_halvings = ctx->width_halvings;
    plan_out->height_halvings = ctx->height_halvings;
    plan_out->span_mul_x = ctx->span_mul_x;
    plan_out->span_mul_y = ctx->span_mul_y;

    smol_scale_destroy (ctx);
}

/* Copies the reference's precalc arrays (uint16 pairs) for one axis.
 * axis 0 = x (relative offsets / whole-pixel counts), 1 = y (absolute).
 * Returns the number of uint16 entries written (<= max_u16). */
uint32_t
ref_probe_precalc (uint32_t w_in, uint32_t h_in, uint32_t w_out, uint32_t h_out,
                   uint32_t axis, uint16_t *dest, uint32_t max_u16)
{
    static char dummy_in [16], dummy_out [16];
    SmolScaleCtx *ctx;
    uint32_t n = 0, i;
    const uint16_t *src;

    ctx = smol_scale_new (dummy_in, SMOL_PIXEL_RGBA8_PREMULTIPLIED, w_in, h_in, w_in * 4,
                          dummy_out, SMOL_PIXEL_RGBA8_PREMULTIPLIED, w_out, h_out, w_out * 4, 0);

    if (axis == 0)
    {
        src = ctx->precalc_x;
        if (ctx->filter_h == SMOL_FILTER_BOX)
            n = (ctx->width_out + 1) * 2;
        else if (ctx->filter_h >= SMOL_FILTER_BILINEAR_0H && ctx->filter_h <= SMOL_FILTER_BILINEAR_6H)
            n = ctx->width_bilin_out * 2;
    }
    else
    {
        src = ctx->precalc_y;
        if (ctx->filter_v == SMOL_FILTER_BOX)
            n = (ctx->height_out + 1) * 2;
        else if (ctx->filter_v >= SMOL_FILTER_BILINEAR_0H && ctx->filter_v <= SMOL_FILTER_BILINEAR_6H)
            n = ctx->height_bilin_out * 2;
    }

    if (n > max_u16)
        n = max_u16;
    for (i = 0; i < n; i++)
        dest [i] = src [i];

    smol_scale_destroy (ctx);
    return n;
}

/* LUT access for the fixture generator (smolscale-private.h:314-328). */
const uint16_t *ref_probe_from_srgb_lut (void) { return _smol_from_srgb_lut; }
const uint8_t  *ref_probe_to_srgb_lut (void)   { return _smol_to_srgb_lut; }
const uint32_t *ref_probe_inv_div_lut (int which)
{
    switch (which)
    {
        case 0: return _smol_inv_div_p8_lut;
        case 1: return _smol_inv_div_p8l_lut;
        case 2: return _smol_inv_div_p16_lut;
        default: return _smol_inv_div_p16l_lut;
    }
}

/* ---- multi-threaded timing driver for bench.py's cpu_baseline leg ------------
 * The reference's own MT pattern (test.c:838-883): one context, n_threads workers,
 * worker i produces output rows [i*R, i*R+R) with R = ceil(h_out / n_threads)
 * through smol_scale_batch(). Here the workers are plain pthreads that each
 * repeat their strip n_frames times (no per-frame pool setup, which favours the
 * CPU relative to test.c, whose timing includes creating the pool). Returns wall
 * seconds for all frames. */

#include <pthread.h>
#include <time.h>

typedef struct
{
    SmolScaleCtx *ctx;
    uint32_t first, n, n_frames;
}
RefMtWorker;

static void *
ref_mt_worker (void *arg)
{
    RefMtWorker *w = arg;
    uint32_t f;

    for (f = 0; f < w->n_frames; f++)
        smol_scale_batch (w->ctx, w->first, w->n);
    return NULL;
}

double
ref_probe_mt_time (const void *pixels_in, uint32_t type_in, uint32_t w_in, uint32_t h_in, uint32_t stride_in,
                   void *pixels_out, uint32_t type_out, uint32_t w_out, uint32_t h_out, uint32_t stride_out,
                   uint32_t with_srgb, uint32_t n_threads, uint32_t n_frames)
{
    SmolScaleCtx *ctx;
    pthread_t *tids;
    RefMtWorker *workers;
    uint32_t rows, i, n_used = 0;
    struct timespec t0, t1;

    if (n_threads < 1)
        n_threads = 1;
    ctx = smol_scale_new (pixels_in, (SmolPixelType) type_in, w_in, h_in, stride_in,
                          pixels_out, (SmolPixelType) type_out, w_out, h_out, stride_out,
                          (uint8_t) with_srgb);
    tids = calloc (n_threads, sizeof (*tids));
    workers = calloc (n_threads, sizeof (*workers));
    rows = (h_out + n_threads - 1) / n_threads;

    clock_gettime (CLOCK_MONOTONIC, &t0);
    for (i = 0; i < n_threads; i++)
    {
        uint32_t first = i * rows;

        if (first >= h_out)
            break;
        workers [i].ctx = ctx;
        workers [i].first = first;
        workers [i].n = (h_out - first < rows) ? h_out - first : rows;
        workers [i].n_frames = n_frames;
        pthread_create (&tids [i], NULL, ref_mt_worker, &workers [i]);
        n_used++;
    }
    for (i = 0; i < n_used; i++)
        pthread_join (tids [i], NULL);
    clock_gettime (CLOCK_MONOTONIC, &t1);

    free (tids);
    free (workers);
    smol_scale_destroy (ctx);
    return (double) (t1.tv_sec - t0.tv_sec) + (double) (t1.tv_nsec - t0.tv_nsec) * 1e-9;
}
