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
 * Backends, preferred first: HIP (gfx950) and a from-scratch CPU implementation
 * (smol_cpu.c). Which ones a context may use is the caller's explicit choice:
 *   SMOL_BACKEND unset or "hip"  HIP only. If it cannot run, the library reports why on
 *                                stderr and abort()s -- the reference's own reaction to an
 *                                unusable configuration (smolscale.c:779-780, 812-813).
 *                                Nothing falls back silently.
 *   SMOL_BACKEND=cpu             the CPU backend only.
 *   SMOL_BACKEND=auto            HIP, else the CPU backend with one line on stderr saying so
 *                                (the reference's "first capable implementation wins",
 *                                smolscale.c:790-810). */

#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smolscale.h"
#include "smol_hip.h"
#include "smol_cpu.h"

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
    SmolHipPlan *plans [SMOL_MAX_DEVICES];  /* one per device (row sharding)      */
    int n_plans;
    SmolHipJob job;                 /* CPU backend: the planned job                */
    SmolCpuPlan *cpu_plan;          /* ... and its tap tables, built once          */
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
 * concurrently, each through its device's side streams -- rows are independent, so
 * there is no exchange between devices. A device may be listed more than once.
 * Without it, SMOL_HIP_DEVICE=<n> (default 0) selects a single GPU.
 * Returns the number of devices, or a negative error for a malformed list. */
static int
parse_device (const char **p, int count, int *out)
{
    char *end;
    long v;

    v = strtol (*p, &end, 10);
    if (end == *p || v < 0 || v >= count)
        return SMOL_HIP_E_INVALID;
    *out = (int) v;
    *p = end;
    return 0;
}

static int
hip_devices_from_env (int *devices)
{
    const char *e = getenv ("SMOL_HIP_DEVICES");
    int count = smol_hip_device_count ();
    int n = 0;

    if (count < 1)
        return SMOL_HIP_E_NO_DEVICE;
    if (count > SMOL_MAX_DEVICES)
        count = SMOL_MAX_DEVICES;

    if (e && *e)
    {
        if (!strcmp (e, "all"))
        {
            for (n = 0; n < count; n++)
                devices [n] = n;
            return n;
        }
        while (*e)
        {
            if (n == SMOL_MAX_DEVICES || parse_device (&e, count, &devices [n]))
                return SMOL_HIP_E_INVALID;
            n++;
            if (*e == ',')
                e++;
            else if (*e)
                return SMOL_HIP_E_INVALID;
        }
        return n ? n : SMOL_HIP_E_INVALID;
    }

    devices [0] = 0;
    e = getenv ("SMOL_HIP_DEVICE");
    if (e && *e && (parse_device (&e, count, &devices [0]) || *e))
        return SMOL_HIP_E_INVALID;
    return 1;
}

static void
hip_backend_fini (SmolScaleCtx *ctx)
{
    int i;

    for (i = 0; i < ctx->n_plans; i++)
        smol_hip_plan_destroy (ctx->plans [i]);
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
    if (n < 0)
        return n;
    ctx->n_plans = 0;
    for (i = 0; i < n; i++)
    {
        rc = smol_hip_plan_create (&job, devices [i], &ctx->plans [i]);
        if (rc)
        {
            hip_backend_fini (ctx);
            return rc;
        }
        ctx->n_plans = i + 1;
    }
    return 0;
}

/* One strip per device, all in flight together; the shim fans them out (asynchronous
 * streams for page-locked memory, its persistent worker threads otherwise). */
static int
hip_backend_batch (const SmolScaleCtx *ctx, void *dest, uint32_t first, uint32_t n)
{
    return smol_hip_scale_rows_host_multi ((SmolHipPlan *const *) ctx->plans, (uint32_t) ctx->n_plans,
                                           ctx->pixels_in, ctx->rowstride_in,
                                           dest, ctx->rowstride_out, first, n);
}

static const SmolBackend hip_backend =
{
    "hip-gfx950", hip_backend_init, hip_backend_batch, hip_backend_fini
};

/* ---- CPU backend (smol_cpu.c) --------------------------------------------------- */

static int
cpu_backend_init (SmolScaleCtx *ctx)
{
    int rc = smol_hip_plan_job (ctx->pixel_type_in, ctx->width_in, ctx->height_in,
                                ctx->pixel_type_out, ctx->width_out, ctx->height_out,
                                ctx->with_srgb, &ctx->job);
    if (rc)
        return rc;
    return smol_cpu_plan_create (&ctx->job, &ctx->cpu_plan);
}

static int
cpu_backend_batch (const SmolScaleCtx *ctx, void *dest, uint32_t first, uint32_t n)
{
    return smol_cpu_plan_scale_rows (ctx->cpu_plan, ctx->pixels_in, ctx->rowstride_in,
                                     dest, ctx->rowstride_out, first, n);
}

static void
cpu_backend_fini (SmolScaleCtx *ctx)
{
    smol_cpu_plan_destroy (ctx->cpu_plan);
    ctx->cpu_plan = NULL;
}

static const SmolBackend cpu_backend =
{
    "cpu", cpu_backend_init, cpu_backend_batch, cpu_backend_fini
};

/* Preferred first, NULL-terminated: the lists SMOL_BACKEND selects from. */
static const SmolBackend *const backends_hip [] = { &hip_backend, NULL };
static const SmolBackend *const backends_cpu [] = { &cpu_backend, NULL };
static const SmolBackend *const backends_auto [] = { &hip_backend, &cpu_backend, NULL };

static const SmolBackend *const *
backends_from_env (void)
{
    const char *e = getenv ("SMOL_BACKEND");

    if (!e || !*e || !strcmp (e, "hip"))
        return backends_hip;
    if (!strcmp (e, "cpu"))
        return backends_cpu;
    if (!strcmp (e, "auto"))
        return backends_auto;
    fprintf (stderr, "smolscale_amd: SMOL_BACKEND=%s is not one of hip, cpu, auto\n", e);
    abort ();
}

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
    const SmolBackend *const *backends = backends_from_env ();
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
            if (i > 0)
            {
                /* never silent: say once per process that a later backend took over */
                static atomic_flag said = ATOMIC_FLAG_INIT;
                if (!atomic_flag_test_and_set (&said))
                {
                    fprintf (stderr, "smolscale_amd: backend %s is not usable here, using %s (SMOL_BACKEND=auto)\n",
                             backends [0]->name, backends [i]->name);
                }
            }
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
