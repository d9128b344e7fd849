/* smol_k_misc.hip -- kernel instantiations + launch wrappers of the k_general / k_wavepx / k_wtile family (gfx950).
 * One translation unit per kernel family so that `make -j' compiles them side by
 * side; the runtime (smol_rt_*.hip) only sees smol_launch.h. */

#include <hip/hip_runtime.h>

#include "smol_kernels.hip.h"
#include "smol_kernel_tile.hip.h"
#include "smol_kernel_wtile.hip.h"
#include "smol_launch.h"

using smolhip::JobDev;

namespace {

} /* namespace */

namespace smolhip {
namespace launch {

void
wtile (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    if (jobs_dev)
        hipLaunchKernelGGL (smolhip::k_wtile_batch, grid, dim3 (64), lds, s, jobs_dev);
    else
        hipLaunchKernelGGL (smolhip::k_wtile_one, grid, dim3 (64), lds, s, rep);
}

void
wavepx (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, bool wide, hipStream_t s)
{
    if (wide)
    {
        if (jobs_dev)
            hipLaunchKernelGGL (smolhip::k_wavepx_batch<true>, grid, dim3 (64), 0, s, jobs_dev);
        else
            hipLaunchKernelGGL (smolhip::k_wavepx_one<true>, grid, dim3 (64), 0, s, rep);
    }
    else
    {
        if (jobs_dev)
            hipLaunchKernelGGL (smolhip::k_wavepx_batch<false>, grid, dim3 (64), 0, s, jobs_dev);
        else
            hipLaunchKernelGGL (smolhip::k_wavepx_one<false>, grid, dim3 (64), 0, s, rep);
    }
}

void
general (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, bool wide, hipStream_t s)
{
    const dim3 block (smolhip::GENERAL_BLOCK);

    if (wide)
    {
        if (jobs_dev)
            hipLaunchKernelGGL (smolhip::k_general_batch<true>, grid, block, 0, s, jobs_dev);
        else
            hipLaunchKernelGGL (smolhip::k_general_one<true>, grid, block, 0, s, rep);
    }
    else
    {
        if (jobs_dev)
            hipLaunchKernelGGL (smolhip::k_general_batch<false>, grid, block, 0, s, jobs_dev);
        else
            hipLaunchKernelGGL (smolhip::k_general_one<false>, grid, block, 0, s, rep);
    }
}

} /* namespace launch */
} /* namespace smolhip */
