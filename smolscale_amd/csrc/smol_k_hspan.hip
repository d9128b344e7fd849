/* smol_k_hspan.hip -- kernel instantiations + launch wrapper of the k_hspan family (row spans: vertical copy,
 * horizontal bilinear at a nearly power-of-two ratio; BASELINE config 4) for gfx950.
 * One translation unit per kernel family so that `make -j' compiles them side by
 * side; the runtime (smol_rt_*.hip) only sees smol_launch.h. */

#include <hip/hip_runtime.h>

#include "smol_kernels.hip.h"
#include "smol_kernel_tile.hip.h"
#include "smol_kernel_hspan.hip.h"
#include "smol_launch.h"

using smolhip::JobDev;

namespace {

template <int BPP, int HX>
void
launch_hspan_k (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    const dim3 block (64 * smolhip::HSPAN_WAVES);
    if (jobs_dev)
        hipLaunchKernelGGL ((smolhip::k_hspan_batch<BPP, HX>), grid, block, lds, s, jobs_dev);
    else
        hipLaunchKernelGGL ((smolhip::k_hspan_one<BPP, HX>), grid, block, lds, s, rep);
}

} /* namespace */

namespace smolhip {
namespace launch {

void
hspan (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    if (rep.bpp_in == 3)
    {
        if (rep.halv_h == 1) launch_hspan_k<3, 1> (jobs_dev, rep, grid, lds, s);
        else                 launch_hspan_k<3, 2> (jobs_dev, rep, grid, lds, s);
    }
    else
    {
        if (rep.halv_h == 0)      launch_hspan_k<4, 0> (jobs_dev, rep, grid, lds, s);
        else if (rep.halv_h == 1) launch_hspan_k<4, 1> (jobs_dev, rep, grid, lds, s);
        else                      launch_hspan_k<4, 2> (jobs_dev, rep, grid, lds, s);
    }
}

} /* namespace launch */
} /* namespace smolhip */
