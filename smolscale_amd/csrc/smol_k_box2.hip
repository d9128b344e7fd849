/* smol_k_box2.hip -- kernel instantiations + launch wrappers of k_boxh / k_boxv, the
 * two-pass box filter (gfx950). One translation unit per kernel family so that
 * `make -j' compiles them side by side; the runtime (smol_rt_*.hip) only sees smol_launch.h. */

#include <hip/hip_runtime.h>

#include "smol_kernels.hip.h"
#include "smol_kernel_tile.hip.h"
#include "smol_kernel_up.hip.h"
#include "smol_kernel_box2.hip.h"
#include "smol_launch.h"

using smolhip::JobDev;

namespace smolhip {
namespace launch {

/* h_grid = pass-1 grid (box2_h_grid: x | y << 16, 4 waves = 4 column tiles per workgroup); v_tiles counts
 * pass-2 waves, a workgroup runs 4 of them */
void
box2 (const JobDev *jobs_dev, const JobDev &rep, uint32_t n_jobs, uint32_t h_grid, uint32_t v_tiles,
      uint32_t h_lds, hipStream_t s)
{
    const dim3 gh (h_grid & 0xffffu, h_grid >> 16, n_jobs), gv ((v_tiles + 3) / 4, n_jobs);
    const uint32_t v_lds = rep.wide ? (uint32_t) ((LUT_LDS_BYTES + 15u) & ~15u) : 0u;

    if (rep.wide)
    {
        if (jobs_dev)
        {
            hipLaunchKernelGGL (smolhip::k_boxh_batch<true>, gh, dim3 (256), h_lds, s, jobs_dev);
            hipLaunchKernelGGL (smolhip::k_boxv_batch<true>, gv, dim3 (256), v_lds, s, jobs_dev);
        }
        else
        {
            hipLaunchKernelGGL (smolhip::k_boxh_one<true>, gh, dim3 (256), h_lds, s, rep);
            hipLaunchKernelGGL (smolhip::k_boxv_one<true>, gv, dim3 (256), v_lds, s, rep);
        }
    }
    else
    {
        if (jobs_dev)
        {
            hipLaunchKernelGGL (smolhip::k_boxh_batch<false>, gh, dim3 (256), h_lds, s, jobs_dev);
            hipLaunchKernelGGL (smolhip::k_boxv_batch<false>, gv, dim3 (256), v_lds, s, jobs_dev);
        }
        else
        {
            hipLaunchKernelGGL (smolhip::k_boxh_one<false>, gh, dim3 (256), h_lds, s, rep);
            hipLaunchKernelGGL (smolhip::k_boxv_one<false>, gv, dim3 (256), v_lds, s, rep);
        }
    }
}

} /* namespace launch */
} /* namespace smolhip */
