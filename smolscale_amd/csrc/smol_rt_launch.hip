/* smol_rt_launch.hip -- from a device job descriptor to a kernel launch (host side; the kernels themselves
 * live behind smol_launch.h, one translation unit per family). */

#include "smol_rt.h"
#include "smol_launch.h"

using namespace smolhip::dispatch;

namespace smolrt {

/* Launch one kernel over `n_jobs' jobs that all share class k and the per-class
 * grouping keys of smol_hip_batch_create. max_* are maxima over the group. */
hipError_t
launch_class (KClass k, const JobDev *jobs_dev, uint32_t n_jobs, const JobDev &rep,
              uint32_t max_w_out, uint32_t max_rows, hipStream_t s, uint32_t max_lds)
{
    namespace L = smolhip::launch;

    if (k == KC_POW2)
    {
        const uint32_t rx = 2u << rep.halv_h;
        const uint32_t opt = rx >= 4 ? 1 : 4 / rx;
        const uint32_t tcols = max_w_out / opt;
        const uint32_t by = pow2_block_rows (rep);
        L::pow2 (jobs_dev, rep, dim3 ((tcols + 63) / 64, (max_rows + by - 1) / by, n_jobs), by, s);
    }
    else if (k == KC_POW2X)
    {
        /* max_w_out / max_rows carry the group's largest column / row tile counts (batches) */
        L::pow2x (jobs_dev, rep, dim3 (jobs_dev ? max_w_out : pow2x_tiles_x (rep),
                                       jobs_dev ? max_rows : pow2x_tiles_y (rep), n_jobs), s);
    }
    else if (k == KC_DOWN)
    {
        /* max_lds carries the largest tile count, max_w_out (batches only) the largest LDS size */
        L::down (jobs_dev, rep, dim3 (max_lds ? max_lds : down_blocks (rep), n_jobs),
                 jobs_dev ? max_w_out : down_lds_bytes (rep), s);
    }
    else if (k == KC_UP)
    {
        L::up (jobs_dev, rep, dim3 (max_lds ? max_lds : up_blocks (rep), n_jobs),
               jobs_dev ? max_w_out : up_lds_bytes (rep), s);
    }
    else if (k == KC_BOX2)
    {
        /* rep.tmp / jobs [i].tmp point at the intermediate images; max_w_out carries the group's largest
         * pass-1 grid (box2_h_grid: x | y << 16), max_rows its largest pass-2 wave count, max_lds the largest LDS size */
        L::box2 (jobs_dev, rep, n_jobs, jobs_dev ? max_w_out : box2_h_grid (rep),
                 jobs_dev ? max_rows : box2_v_tiles (rep), jobs_dev ? max_lds : box2_h_lds_bytes (rep), s);
    }
    else if (k == KC_HSPAN)
    {
        /* max_lds carries the group's largest block count; groups share pixel size and halvings */
        L::hspan (jobs_dev, rep, dim3 (max_lds ? max_lds : hspan_blocks (rep), n_jobs), hspan_lds_bytes (rep), s);
    }
    else if (k == KC_TILE || k == KC_WTILE)
    {
        /* groups share tile_rows and LDS size */
        const uint32_t tw = 64u * rep.tile_cpl / rep.tile_lpc;
        const uint32_t tiles_x = (max_w_out + tw - 1) / tw;
        const uint32_t tiles_y = (max_rows + rep.tile_rows - 1) / rep.tile_rows;
        /* k_tile: (column tiles, row tiles, jobs); k_wtile: (tiles, jobs) */
        const dim3 grid = (k == KC_TILE) ? dim3 (tiles_x, tiles_y, n_jobs) : dim3 (tiles_x * tiles_y, n_jobs);
        const uint32_t lds = max_lds ? max_lds : tile_lds_bytes (rep);

        if (k == KC_WTILE)
            L::wtile (jobs_dev, rep, grid, lds, s);
        else
            L::tile (jobs_dev, rep, grid, lds, s);
    }
    else if (k == KC_WAVEPX || k == KC_WAVEPX_WIDE)
    {
        /* groups hold jobs of one output width */
        L::wavepx (jobs_dev, rep, dim3 (rep.w_out * max_rows, n_jobs), k == KC_WAVEPX_WIDE, s);
    }
    else
    {
        L::general (jobs_dev, rep, dim3 (((max_w_out + 255u) / 256u) * max_rows, n_jobs), k == KC_GENERAL_WIDE, s);
    }
    return hipGetLastError ();
}

hipError_t
launch_single (JobDev &jd, hipStream_t s, void *scratch, size_t scratch_bytes, KClass *k_out)
{
    const KClass k = classify (jd);

    if (k_out)
        *k_out = k;
    if (k == KC_BOX2)
    {
        /* the intermediate image must hold what THIS call's classification needs (a caller that sized it from
         * another probe and got it wrong is refused here instead of writing past the buffer) */
        if (!scratch || scratch_bytes < box2_tmp_bytes (jd))
            return hipErrorInvalidValue;
        jd.tmp = static_cast<uint8_t *> (scratch);
    }
    return launch_class (k, nullptr, 1, jd, jd.w_out, jd.n_rows, s);
}

} /* namespace smolrt */

/* Which kernel family takes these rows: the dispatcher's own decision (fill_jobdev + classify), no GPU involved.
 * For the histograms of tools/fuzz/gpu_fuzz.py, tests/test_hip_sweep.py and tools/kernel_histogram.py. */
extern "C" const char *
smol_hip_kernel_name (const SmolHipJob *job, const void *src, uint32_t rowstride_in,
                      const void *dst, uint32_t rowstride_out, uint32_t first_row, uint32_t n_rows)
{
    if (!job || job->pixel_type_in >= 10 || job->pixel_type_out >= 10 || !n_rows)
        return "";
    smolhip::JobDev jd;
    fill_jobdev (jd, *job, src, rowstride_in, const_cast<void *> (dst), rowstride_out, first_row, n_rows);
    const KClass k = classify (jd);
    return kclass_name (k, jd);
}
