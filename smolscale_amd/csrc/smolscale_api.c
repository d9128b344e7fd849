/* smolscale_api.c -- public smol_scale_*() entry points (include/smolscale.h)
 * over an ordered table of backends.
 *
 * Mirrors the shape of the reference's dispatcher: a context filled at creation
 * (reference smol_scale_init(), smolscale.c:816-870), an ordered list of
 * implementations with the preferred one first (get_implementations(),
 * smolscale.c:725-814) and an outer loop over output rows (do_rows(),
 * smolscale.c:507-546). What differs is the grain of the backend interface: the
 * reference installs per-ROW function pointers (SmolImplementation,
 * smolscale-private.h:270-278); a GPU backend takes a whole RANGE of rows per
 * call (`batch_func'), so the row loop lives behind the seam, on the device.
 *
 * There is exactly one backend in this library, HIP. If it cannot run, the
 * library reports why on stderr and abort()s -- the reference's own reaction to
 * an unusable configuration (smolscale.c:779-780, 812-813). No CPU fallback. */

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smolscale.h"
#include "smol_hip.h"

typedef struct SmolBackend SmolBackend;

#define SMOL_MAX_DEVICES 16

struct SmolScaleCtx
{
    const char *pixels_in;
    char *pixels_out;
    uint32_t width_in, height_in, rowstride_in;
    uint32_t width_out, height_out, rowstride_out;
    SmolPixelType pixel_type_in, pixel_type_out;
    uint8_t with_srgb;

    SmolPostRowFunc *post_row_func;
    void *user_data;

    const SmolBackend *backend;
    void *backend_state;            /* SmolHipPlan * of the first device        */
    void *more_plans [SMOL_MAX_DEVICES - 1];   /* further devices (row sharding) */
    int n_plans;
};

/* The seam. init() plans the job and acquires backend resources; batch() must
 * produce rows [first, first + n) at dest (row `first' at offset 0, rowstride_out
 * apart) and may be entered concurrently for disjoint ranges; fini() releases. */
struct SmolBackend
{
    const char *name;
    int (*init) (SmolScaleCtx *ctx);
    int (*batch) (const SmolScaleCtx *ctx, void *dest, uint32_t first, uint32_t n);
    void (*fini) (SmolScaleCtx *ctx);
};

/* ---- HIP backend ------------------------------------------------------------ */

/* Which GPUs a context uses. SMOL_HIP_DEVICES="0,1,2,3" or "all": output rows of
 * every batch are split into one strip per listed device (the reference's own
 * strip rule for its worker threads, test.c:870-877) and the strips run
 * concurrently, each on its device's stream -- rows are independent, so there is
 * no exchange between devices. A device may be listed more than once.
 * Without it, SMOL_HIP_DEVICE=<n> (default 0) selects a single GPU. */
static int
hip_devices_from_env (int *devices)
{
    const char *e = getenv ("SMOL_HIP_DEVICES");
    int n = 0;

    if (e && *e)
    {
        if (!strcmp (e, "all"))
        {
            int count = smol_hip_device_count ();
            for (n = 0; n < count && n < SMOL_MAX_DEVICES; n++)
                devices [n] = n;
        }
        else
        {
            const char *p = e;
            while (*p && n < SMOL_MAX_DEVICES)
            {
                devices [n++] = atoi (p);
                while (*p && *p != ',')
                    p++;
                if (*p == ',')
                    p++;
            }
        }
    }

    if (n == 0)
    {
        e = getenv ("SMOL_HIP_DEVICE");
        devices [0] = e ? atoi (e) : 0;
        n = 1;
    }
    return n;
}

static void
hip_backend_fini (SmolScaleCtx *ctx)
{
    int i;

    if (ctx->backend_state)
        smol_hip_plan_destroy ((SmolHipPlan *) ctx->backend_state);
    for (i = 0; i + 1 < ctx->n_plans; i++)
        smol_hip_plan_destroy ((SmolHipPlan *) ctx->more_plans [i]);
    ctx->backend_state = NULL;
    ctx->n_plans = 0;
}

static int
hip_backend_init (SmolScaleCtx *ctx)
{
    SmolHipJob job;
    int devices [SMOL_MAX_DEVICES];
    int n, i, rc;

    rc = smol_hip_plan_job (ctx->pixel_type_in, ctx->width_in, ctx->height_in,
                            ctx->pixel_type_out, ctx->width_out, ctx->height_out,
                            ctx->with_srgb, &job);
    if (rc)
        return rc;

    n = hip_devices_from_env (devices);
    ctx->n_plans = 0;
    for (i = 0; i < n; i++)
    {
        SmolHipPlan *plan = NULL;

        rc = smol_hip_plan_create (&job, devices [i], &plan);
        if (rc)
        {
            hip_backend_fini (ctx);
            return rc;
        }
        if (i == 0)
            ctx->backend_state = plan;
        else
            ctx->more_plans [i - 1] = plan;
        ctx->n_plans = i + 1;
    }
    return 0;
}

typedef struct
{
    const SmolScaleCtx *ctx;
    SmolHipPlan *plan;
    void *dest;
    uint32_t first, n;
    int rc;
}
HipStrip;

static void *
hip_strip_worker (void *arg)
{
    HipStrip *s = arg;

    s->rc = smol_hip_scale_rows_host (s->plan, s->ctx->pixels_in, s->ctx->rowstride_in,
                                      s->dest, s->ctx->rowstride_out, s->first, s->n);
    return NULL;
}

static int
hip_backend_batch (const SmolScaleCtx *ctx, void *dest, uint32_t first, uint32_t n)
{
    HipStrip strips [SMOL_MAX_DEVICES];
    pthread_t tids [SMOL_MAX_DEVICES];
    uint32_t rows, done = 0;
    int n_strips = 0, i, rc = 0;

    if (ctx->n_plans <= 1 || n < (uint32_t) ctx->n_plans)
        return smol_hip_scale_rows_host ((SmolHipPlan *) ctx->backend_state,
                                         ctx->pixels_in, ctx->rowstride_in,
                                         dest, ctx->rowstride_out, first, n);

    rows = (n + (uint32_t) ctx->n_plans - 1) / (uint32_t) ctx->n_plans;
    for (i = 0; i < ctx->n_plans && done < n; i++)
    {
        HipStrip *s = &strips [n_strips];

        s->ctx = ctx;
        s->plan = (SmolHipPlan *) (i == 0 ? ctx->backend_state : ctx->more_plans [i - 1]);
        s->first = first + done;
        s->n = (n - done < rows) ? n - done : rows;
        s->dest = (char *) dest + (size_t) ctx->rowstride_out * done;
        s->rc = 0;
        done += s->n;
        if (pthread_create (&tids [n_strips], NULL, hip_strip_worker, s) != 0)
        {
            hip_strip_worker (s);          /* no thread available: run it here */
            tids [n_strips] = (pthread_t) 0;
            if (s->rc && !rc)
                rc = s->rc;
            continue;
        }
        n_strips++;
    }
    for (i = 0; i < n_strips; i++)
    {
        pthread_join (tids [i], NULL);
        if (strips [i].rc && !rc)
            rc = strips [i].rc;
    }
    return rc;
}

static const SmolBackend hip_backend =
{
    "hip-gfx950", hip_backend_init, hip_backend_batch, hip_backend_fini
};

/* Preferred first, NULL-terminated. */
static const SmolBackend *const backends [] = { &hip_backend, NULL };

/* ---- dispatcher -------------------------------------------------------------- */

static void
die (const char *what, const char *backend, int rc)
{
    fprintf (stderr, "smolscale_amd: %s failed in backend %s: %s (%d)\n",
             what, backend, smol_hip_strerror (rc), rc);
    abort ();
}

static void
ctx_init (SmolScaleCtx *ctx,
          const void *pixels_in, SmolPixelType pixel_type_in,
          uint32_t width_in, uint32_t height_in, uint32_t rowstride_in,
          void *pixels_out, SmolPixelType pixel_type_out,
          uint32_t width_out, uint32_t height_out, uint32_t rowstride_out,
          uint8_t with_srgb, SmolPostRowFunc post_row_func, void *user_data)
{
    int i, rc = SMOL_HIP_E_NO_DEVICE;

    memset (ctx, 0, sizeof (*ctx));
    ctx->pixels_in = pixels_in;
    ctx->pixels_out = pixels_out;
    ctx->width_in = width_in;
    ctx->height_in = height_in;
    ctx->rowstride_in = rowstride_in;
    ctx->width_out = width_out;
    ctx->height_out = height_out;
    ctx->rowstride_out = rowstride_out;
    ctx->pixel_type_in = pixel_type_in;
    ctx->pixel_type_out = pixel_type_out;
    ctx->with_srgb = with_srgb;
    ctx->post_row_func = post_row_func;
    ctx->user_data = user_data;

    for (i = 0; backends [i]; i++)
    {
        rc = backends [i]->init (ctx);
        if (rc == 0)
        {
            ctx->backend = backends [i];
            return;
        }
    }

    die ("context creation", backends [0]->name, rc);
}

/* Counterpart of the reference's do_rows() + scale_outrow() (smolscale.c:492-546). */
static void
produce_rows (const SmolScaleCtx *ctx, void *dest, uint32_t first, uint32_t n)
{
    uint32_t i;
    int rc;

    if (n == 0)
        return;

    rc = ctx->backend->batch (ctx, dest, first, n);
    if (rc)
        die ("row batch", ctx->backend->name, rc);

    if (ctx->post_row_func)
    {
        for (i = 0; i < n; i++)
            ctx->post_row_func ((uint32_t *) ((char *) dest + (size_t) ctx->rowstride_out * i),
                                (int) ctx->width_out, ctx->user_data);
    }
}

/* ---- public API ---------------------------------------------------------------- */

SmolScaleCtx *
smol_scale_new_full (const void *pixels_in, SmolPixelType pixel_type_in,
                     uint32_t width_in, uint32_t height_in, uint32_t rowstride_in,
                     void *pixels_out, SmolPixelType pixel_type_out,
                     uint32_t width_out, uint32_t height_out, uint32_t rowstride_out,
                     uint8_t with_srgb,
                     SmolPostRowFunc post_row_func, void *user_data)
{
    SmolScaleCtx *ctx = malloc (sizeof (SmolScaleCtx));

    if (!ctx)
        abort ();
    ctx_init (ctx, pixels_in, pixel_type_in, width_in, height_in, rowstride_in,
              pixels_out, pixel_type_out, width_out, height_out, rowstride_out,
              with_srgb, post_row_func, user_data);
    return ctx;
}

SmolScaleCtx *
smol_scale_new (const void *pixels_in, SmolPixelType pixel_type_in,
                uint32_t width_in, uint32_t height_in, uint32_t rowstride_in,
                void *pixels_out, SmolPixelType pixel_type_out,
                uint32_t width_out, uint32_t height_out, uint32_t rowstride_out,
                uint8_t with_srgb)
{
    return smol_scale_new_full (pixels_in, pixel_type_in, width_in, height_in, rowstride_in,
                                pixels_out, pixel_type_out, width_out, height_out, rowstride_out,
                                with_srgb, NULL, NULL);
}

void
smol_scale_destroy (SmolScaleCtx *scale_ctx)
{
    if (!scale_ctx)
        return;
    scale_ctx->backend->fini (scale_ctx);
    free (scale_ctx);
}

void
smol_scale_simple (const void *pixels_in, SmolPixelType pixel_type_in,
                   uint32_t width_in, uint32_t height_in, uint32_t rowstride_in,
                   void *pixels_out, SmolPixelType pixel_type_out,
                   uint32_t width_out, uint32_t height_out, uint32_t rowstride_out,
                   uint8_t with_srgb)
{
    SmolScaleCtx ctx;

    ctx_init (&ctx, pixels_in, pixel_type_in, width_in, height_in, rowstride_in,
              pixels_out, pixel_type_out, width_out, height_out, rowstride_out,
              with_srgb, NULL, NULL);
    produce_rows (&ctx, ctx.pixels_out, 0, ctx.height_out);
    ctx.backend->fini (&ctx);
}

void
smol_scale_batch (const SmolScaleCtx *scale_ctx, uint32_t first_outrow, uint32_t n_outrows)
{
    produce_rows (scale_ctx,
                  scale_ctx->pixels_out + (size_t) scale_ctx->rowstride_out * first_outrow,
                  first_outrow, n_outrows);
}

void
smol_scale_batch_full (const SmolScaleCtx *scale_ctx, void *outrows_dest,
                       uint32_t first_outrow, uint32_t n_outrows)
{
    produce_rows (scale_ctx, outrows_dest, first_outrow, n_outrows);
}
