/* smolscale-hip.c -- the reference-side binding of the MI355X backend: the one file a
 * smolscale maintainer adds next to smolscale-avx2.c. It fills the `batch_func' slot this
 * integration adds to SmolImplementation (smolscale-private.h) with a call into the C-ABI of
 * include/smol_hip.h; nothing of HIP or C++ crosses into the reference's C.
 *
 * Contract with smolscale.c (see tests/integration/apply_hip_binding.py for the hunks):
 *   - get_implementations() asks _smol_get_hip_implementation() last; init_batch_func()
 *     returns 0 when no GPU can take the job, and the CPU row functions installed before
 *     stay in charge (silent fall-through; SMOL_HIP_VERBOSE=1 reports the choice once).
 *   - do_rows() hands whole row ranges to batch_func() when scale_ctx->batch_func is set.
 *   - smol_scale_finalize() calls fini_batch_func().
 *
 * This file is part of the test infrastructure of smolscale_amd (tests/integration): it is
 * compiled into a scratch copy of the reference by build_ref_with_hip.sh, never into the
 * product library. */

#include <stdio.h>
#include <stdlib.h>

#include "smolscale-private.h"
#include "smol_hip.h"

static int
hip_batch (const SmolScaleCtx *scale_ctx, void *outrows_dest, uint32_t first_row, uint32_t n_rows)
{
    return smol_hip_scale_rows_host ((SmolHipPlan *) scale_ctx->hip_plan,
                                     scale_ctx->pixels_in, scale_ctx->rowstride_in,
                                     outrows_dest, scale_ctx->rowstride_out,
                                     first_row, n_rows);
}

static void
report_once (const char *what)
{
    static int said;
    const char *e = getenv ("SMOL_HIP_VERBOSE");

    if (e && *e && *e != '0' && !said)
    {
        said = 1;
        fprintf (stderr, "smolscale: %s\n", what);
    }
}

/* 1 = this context will be served by the GPU */
static int
hip_init_batch (SmolScaleCtx *scale_ctx)
{
    SmolHipJob job;
    SmolHipPlan *plan = NULL;
    const char *e = getenv ("SMOL_HIP_DEVICE");
    int device = e ? atoi (e) : 0;

    scale_ctx->hip_plan = NULL;

    if (smol_hip_device_count () < 1)
    {
        report_once ("no HIP device, using the CPU implementations");
        return 0;
    }
    /* The planner takes the caller's with_srgb; get_implementations() may already have
     * switched gamma off for extreme ratios (smolscale.c:760-770) -- the planner applies
     * the same rule, so either reading gives the same job. */
    if (smol_hip_plan_job (scale_ctx->pixel_type_in, scale_ctx->width_in, scale_ctx->height_in,
                           scale_ctx->pixel_type_out, scale_ctx->width_out, scale_ctx->height_out,
                           scale_ctx->gamma_type == SMOL_GAMMA_SRGB_LINEAR, &job) != 0
        || smol_hip_plan_create (&job, device, &plan) != 0)
    {
        report_once ("HIP backend declined the job, using the CPU implementations");
        return 0;
    }

    scale_ctx->hip_plan = plan;
    report_once ("HIP backend (gfx950) active");
    return 1;
}

static void
hip_fini_batch (SmolScaleCtx *scale_ctx)
{
    if (scale_ctx->hip_plan)
        smol_hip_plan_destroy ((SmolHipPlan *) scale_ctx->hip_plan);
    scale_ctx->hip_plan = NULL;
}

static const SmolImplementation hip_implementation =
{
    .batch_func = hip_batch,
    .init_batch_func = hip_init_batch,
    .fini_batch_func = hip_fini_batch
};

const SmolImplementation *
_smol_get_hip_implementation (void)
{
    return &hip_implementation;
}
