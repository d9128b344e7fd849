/* smol_k_stream.hip -- kernel instantiations + launch wrappers of the k_pow2 / k_pow2x (streaming 2:1, 4:1, 8:1) family (gfx950).
 * One translation unit per kernel family so that `make -j' compiles them side by
 * side; the runtime (smol_rt_*.hip) only sees smol_launch.h. */

#include <hip/hip_runtime.h>

#include "smol_kernels.hip.h"
#include "smol_kernel_tile.hip.h"
#include "smol_kernel_pow2x.hip.h"
#include "smol_launch.h"

using smolhip::JobDev;

namespace {

/* A launch either reads descriptors from device memory (jobs_dev != NULL, one per
 * blockIdx.y) or takes the single descriptor `rep' by value as kernel argument. */
template <int HX, int HY>
void
launch_pow2_one (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t by, hipStream_t s)
{
    /* Source rows that do not start on 256-byte boundaries (a crop of a larger image, an odd width): the 1 KiB a
     * wave instruction reads then shares its first and last 128-byte line with the neighbouring column tiles --
     * deal the workgroups out so that neighbours run on the same XCD and meet in its L2 (smol_jobdev.h: TileSwz).
     * Aligned rows share nothing and keep the plain 3-D grid. */
    const bool shared_lines = smolhip::dispatch::pow2_shared_lines (rep);
    uint32_t n;
    const smolhip::TileSwz swz = smolhip::swizzle_make (grid.x, grid.y, grid.z,
                                                        shared_lines && !smolhip::tune_env ("SMOL_POW2_NO_SWZ"), &n);
    if (n)
        grid = dim3 (n);
    if (jobs_dev)
        hipLaunchKernelGGL ((smolhip::k_pow2_batch<HX, HY>), grid, dim3 (64, by), 0, s, jobs_dev, swz);
    else
        hipLaunchKernelGGL ((smolhip::k_pow2_one<HX, HY>), grid, dim3 (64, by), 0, s, rep, swz);
}

template <int HX>
void
launch_pow2_hy (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t by, hipStream_t s)
{
    switch (rep.halv_v)
    {
        case 0: launch_pow2_one<HX, 0> (jobs_dev, rep, grid, by, s); break;
        case 1: launch_pow2_one<HX, 1> (jobs_dev, rep, grid, by, s); break;
        default: launch_pow2_one<HX, 2> (jobs_dev, rep, grid, by, s); break;
    }
}

/* k_pow2x: one entry point per (halvings, input kind) */
template <int H, int IN>
void
launch_pow2x_k (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, hipStream_t s)
{
    if (jobs_dev)
        hipLaunchKernelGGL ((smolhip::k_pow2x_batch<H, IN>), grid, dim3 (64, 16), 0, s, jobs_dev);
    else
        hipLaunchKernelGGL ((smolhip::k_pow2x_one<H, IN>), grid, dim3 (64, 16), 0, s, rep);
}

template <int H>
void
launch_pow2x_h (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, hipStream_t s)
{
    if (jobs_dev && rep.px_mixed)
    {
        hipLaunchKernelGGL ((smolhip::k_pow2x_mixed_batch<H>), grid, dim3 (64, 16), 0, s, jobs_dev);
        return;
    }
    switch (rep.px_kind)
    {
        case smolhip::PX_N32: launch_pow2x_k<H, smolhip::PX_N32> (jobs_dev, rep, grid, s); break;
        case smolhip::PX_N24: launch_pow2x_k<H, smolhip::PX_N24> (jobs_dev, rep, grid, s); break;
        case smolhip::PX_W32: launch_pow2x_k<H, smolhip::PX_W32> (jobs_dev, rep, grid, s); break;
        case smolhip::PX_W16: launch_pow2x_k<H, smolhip::PX_W16> (jobs_dev, rep, grid, s); break;
        default:              launch_pow2x_k<H, smolhip::PX_W24> (jobs_dev, rep, grid, s); break;
    }
}

} /* namespace */

namespace smolhip {
namespace launch {

void
pow2 (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t by, hipStream_t s)
{
    switch (rep.halv_h)
    {
        case 0: launch_pow2_hy<0> (jobs_dev, rep, grid, by, s); break;
        case 1: launch_pow2_hy<1> (jobs_dev, rep, grid, by, s); break;
        default: launch_pow2_hy<2> (jobs_dev, rep, grid, by, s); break;
    }
}

void
pow2x (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, hipStream_t s)
{
    switch (rep.halv_h)
    {
        case 0: launch_pow2x_h<0> (jobs_dev, rep, grid, s); break;
        case 1: launch_pow2x_h<1> (jobs_dev, rep, grid, s); break;
        default: launch_pow2x_h<2> (jobs_dev, rep, grid, s); break;
    }
}

} /* namespace launch */
} /* namespace smolhip */
