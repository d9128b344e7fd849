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

/* the KIND of a two-pass job (device twin: lin_pass_kind) */
int
lin_pass_kind (const JobDev &J)
{
    if (!J.p16)
        return J.bpp_in == 3 ? 1 : 0;
    if (smolhip::dispatch::twopass_hbox (J))
        return J.linear ? 2 : 3;
    return (!J.linear || J.filter_h == smolhip::F_BOX || J.filter_v == smolhip::F_BOX) ? 3 : 2;
}

template <int HX, int KIND>
void
launch_linh_k (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    if (jobs_dev)
        hipLaunchKernelGGL ((smolhip::k_linh_batch<HX, KIND>), grid, dim3 (256), lds, s, jobs_dev);
    else
        hipLaunchKernelGGL ((smolhip::k_linh_one<HX, KIND>), grid, dim3 (256), lds, s, rep);
}

template <int HX>
void
launch_linh_h (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    switch (lin_pass_kind (rep))
    {
        case 0:  launch_linh_k<HX, 0> (jobs_dev, rep, grid, lds, s); break;
        case 1:  launch_linh_k<HX, 1> (jobs_dev, rep, grid, lds, s); break;
        case 2:  launch_linh_k<HX, 2> (jobs_dev, rep, grid, lds, s); break;
        default: launch_linh_k<HX, 3> (jobs_dev, rep, grid, lds, s); break;
    }
}

/* k_linh: pass 1 of the generic two-pass road (smol_kernel_lind.hip.h), one entry point per (halvings, kind) */
void
launch_linh (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    switch (rep.halv_h)
    {
        case 0:  launch_linh_h<0> (jobs_dev, rep, grid, lds, s); break;
        case 1:  launch_linh_h<1> (jobs_dev, rep, grid, lds, s); break;
        default: launch_linh_h<2> (jobs_dev, rep, grid, lds, s); break;
    }
}

template <int KIND>
void
launch_linv_k (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    if (jobs_dev)
        hipLaunchKernelGGL ((smolhip::k_linv_batch<KIND>), grid, dim3 (256), lds, s, jobs_dev);
    else
        hipLaunchKernelGGL ((smolhip::k_linv_one<KIND>), grid, dim3 (256), lds, s, rep);
}

void
launch_linv (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    switch (lin_pass_kind (rep))
    {
        case 2:  launch_linv_k<2> (jobs_dev, rep, grid, lds, s); break;
        case 3:  launch_linv_k<3> (jobs_dev, rep, grid, lds, s); break;
        default: launch_linv_k<0> (jobs_dev, rep, grid, lds, s); break;
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
    else if (rep.wide && rep.tile_lpc >= 16)
    {
        if (jobs_dev)
            hipLaunchKernelGGL ((smolhip::k_boxh_batch<true, true>), gh, dim3 (256), h_lds, s, jobs_dev);
        else
            hipLaunchKernelGGL ((smolhip::k_boxh_one<true, true>), gh, dim3 (256), h_lds, s, rep);
    }
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
        launch_linv (jobs_dev, rep, gv, v_lds, s);
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
