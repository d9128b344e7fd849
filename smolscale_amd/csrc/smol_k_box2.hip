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

namespace {

template <int HX, bool IN24>
void
launch_linh_k (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    if (jobs_dev)
        hipLaunchKernelGGL ((smolhip::k_linh_batch<HX, IN24>), grid, dim3 (256), lds, s, jobs_dev);
    else
        hipLaunchKernelGGL ((smolhip::k_linh_one<HX, IN24>), grid, dim3 (256), lds, s, rep);
}

/* k_linh: pass 1 of the generic two-pass road (smol_kernel_lind.hip.h), one entry point per (halvings, 24-bit rows) */
void
launch_linh (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    const bool in24 = rep.bpp_in == 3;
    switch (rep.halv_h)
    {
        case 0:  if (in24) launch_linh_k<0, true> (jobs_dev, rep, grid, lds, s); else launch_linh_k<0, false> (jobs_dev, rep, grid, lds, s); break;
        case 1:  if (in24) launch_linh_k<1, true> (jobs_dev, rep, grid, lds, s); else launch_linh_k<1, false> (jobs_dev, rep, grid, lds, s); break;
        default: if (in24) launch_linh_k<2, true> (jobs_dev, rep, grid, lds, s); else launch_linh_k<2, false> (jobs_dev, rep, grid, lds, s); break;
    }
}

} /* namespace */

namespace smolhip {
namespace launch {

/* h_grid = pass-1 grid (box2_h_grid: x | y << 16, 4 waves = 4 column tiles per workgroup); v_tiles counts
 * pass-2 waves, a workgroup runs 4 of them */
void
box2 (const JobDev *jobs_dev, const JobDev &rep, uint32_t n_jobs, uint32_t h_grid, uint32_t v_tiles,
      uint32_t h_lds, hipStream_t s)
{
    using smolhip::dispatch::twopass_lin_h;
    using smolhip::dispatch::twopass_lin_v;
    const dim3 gh (h_grid & 0xffffu, h_grid >> 16, n_jobs), gv ((v_tiles + 3) / 4, n_jobs);
    const uint32_t v_lds = rep.wide ? (uint32_t) ((LUT_LDS_BYTES + 15u) & ~15u) : 0u;

    /* pass 1: horizontal */
    if (twopass_lin_h (rep))
        launch_linh (jobs_dev, rep, gh, h_lds, s);
    else if (rep.wide)
    {
        if (jobs_dev)
            hipLaunchKernelGGL (smolhip::k_boxh_batch<true>, gh, dim3 (256), h_lds, s, jobs_dev);
        else
            hipLaunchKernelGGL (smolhip::k_boxh_one<true>, gh, dim3 (256), h_lds, s, rep);
    }
    else
    {
        if (jobs_dev)
            hipLaunchKernelGGL (smolhip::k_boxh_batch<false>, gh, dim3 (256), h_lds, s, jobs_dev);
        else
            hipLaunchKernelGGL (smolhip::k_boxh_one<false>, gh, dim3 (256), h_lds, s, rep);
    }

    /* pass 2: vertical */
    if (twopass_lin_v (rep))
    {
        if (jobs_dev)
            hipLaunchKernelGGL (smolhip::k_linv_batch<0>, gv, dim3 (256), v_lds, s, jobs_dev);
        else
            hipLaunchKernelGGL (smolhip::k_linv_one<0>, gv, dim3 (256), v_lds, s, rep);
    }
    else if (rep.wide)
    {
        if (jobs_dev)
            hipLaunchKernelGGL (smolhip::k_boxv_batch<true>, gv, dim3 (256), v_lds, s, jobs_dev);
        else
            hipLaunchKernelGGL (smolhip::k_boxv_one<true>, gv, dim3 (256), v_lds, s, rep);
    }
    else
    {
        if (jobs_dev)
            hipLaunchKernelGGL (smolhip::k_boxv_batch<false>, gv, dim3 (256), v_lds, s, jobs_dev);
        else
            hipLaunchKernelGGL (smolhip::k_boxv_one<false>, gv, dim3 (256), v_lds, s, rep);
    }
}

} /* namespace launch */
} /* namespace smolhip */
