/* smol_k_updown.hip -- kernel instantiations + launch wrappers of the k_up / k_lind (magnification; 128bpp-class minification) family (gfx950).
 * One translation unit per kernel family so that `make -j' compiles them side by
 * side; the runtime (smol_rt_*.hip) only sees smol_launch.h. */

#include <hip/hip_runtime.h>

#include "smol_kernels.hip.h"
#include "smol_kernel_tile.hip.h"
#include "smol_kernel_up.hip.h"
#include "smol_kernel_lind.hip.h"
#include "smol_launch.h"

using smolhip::JobDev;

namespace {

/* k_lind: one entry point per (horizontal halvings, input kind: 0 = 32-bit rows, 1 = 24-bit rows, 2 = premul16 + linear) */
template <int HX, int IN>
void
launch_lind_k (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
#ifdef SMOL_TUNING
    /* timing ablations (tuning builds only): the switch word of smol_kernel_lind.hip.h, once per process */
    static const bool dbg_set = [] {
        const char *e = getenv ("SMOL_LIND_DBG");
        const uint32_t v = e ? (uint32_t) atoi (e) : 0u;
        return hipMemcpyToSymbol (HIP_SYMBOL (smolhip::smol_lind_dbg_dev), &v, sizeof v) == hipSuccess;
    } ();
    (void) dbg_set;
#endif
    /* tiles of at most 64 columns (one or two halvings): the one-column flavour, a kernel of its own */
    const dim3 block (64u * smolhip::dispatch::lind_waves (rep));
    if (HX >= 1 && smolhip::dispatch::lind_narrow (rep))
    {
        if (jobs_dev)
            hipLaunchKernelGGL ((smolhip::k_lind_batch<HX, IN, HX >= 1>), grid, block, lds, s, jobs_dev);
        else
            hipLaunchKernelGGL ((smolhip::k_lind_one<HX, IN, HX >= 1>), grid, block, lds, s, rep);
        return;
    }
    /* magnification on both axes into rows that start on 64-byte boundaries: the kernel with streaming stores */
    if (HX == 0 && smolhip::dispatch::lind_nt (rep) && rep.halv_v == 0)     /* (a per-sample step below 1 never comes with a halving) */
    {
        if (jobs_dev)
            hipLaunchKernelGGL ((smolhip::k_lind_batch<0, IN, false, true>), grid, block, lds, s, jobs_dev);
        else
            hipLaunchKernelGGL ((smolhip::k_lind_one<0, IN, false, true>), grid, block, lds, s, rep);
        return;
    }
    /* no halving on either axis (ratios 1-2, magnifications into rows that start anywhere): the vertical sample count compiled in */
    if (HX == 0 && smolhip::dispatch::lind_hv0 (rep))
    {
        if (jobs_dev)
            hipLaunchKernelGGL ((smolhip::k_lind_batch<0, IN, false, false, true>), grid, block, lds, s, jobs_dev);
        else
            hipLaunchKernelGGL ((smolhip::k_lind_one<0, IN, false, false, true>), grid, block, lds, s, rep);
        return;
    }
    if (jobs_dev)
        hipLaunchKernelGGL ((smolhip::k_lind_batch<HX, IN>), grid, block, lds, s, jobs_dev);
    else
        hipLaunchKernelGGL ((smolhip::k_lind_one<HX, IN>), grid, block, lds, s, rep);
}

template <int IN>
void
launch_lind (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    switch (rep.halv_h)
    {
        case 0:  launch_lind_k<0, IN> (jobs_dev, rep, grid, lds, s); break;
        case 1:  launch_lind_k<1, IN> (jobs_dev, rep, grid, lds, s); break;
        default: launch_lind_k<2, IN> (jobs_dev, rep, grid, lds, s); break;
    }
}

/* k_up: one entry point per (input alignment kind, output kind) */
template <int INK, int OUT>
void
launch_up_k (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    if (jobs_dev)
        hipLaunchKernelGGL ((smolhip::k_up_batch<INK, OUT>), grid, dim3 (64), lds, s, jobs_dev);
    else
        hipLaunchKernelGGL ((smolhip::k_up_one<INK, OUT>), grid, dim3 (64), lds, s, rep);
}

template <int INK>
void
launch_up_in (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    if (rep.up_wide == 1)
    {
        launch_up_k<INK, smolhip::OUT_W16> (jobs_dev, rep, grid, lds, s);
        return;
    }
#ifdef SMOL_TUNING
    /* (only the tuning build's SMOL_UP_ALWAYS sends unassociated, 24-bit or byte-aligned output here: up_setup keeps k_up to
     * its fast store form, and the shipped library carries no kernels it never launches) */
    switch (rep.tile_variant & 3u)
    {
        case smolhip::OUT_P4: launch_up_k<INK, smolhip::OUT_P4> (jobs_dev, rep, grid, lds, s); break;
        case smolhip::OUT_U4: launch_up_k<INK, smolhip::OUT_U4> (jobs_dev, rep, grid, lds, s); break;
        default:              launch_up_k<INK, smolhip::OUT_ANY> (jobs_dev, rep, grid, lds, s); break;
    }
#else
    /* (the shipped library: the 16-byte form -- premultiplied or unassociated --, and 24-bit output as twelve bytes per lane) */
    if ((rep.tile_variant & 3u) == smolhip::OUT_ANY)
        launch_up_k<INK, smolhip::OUT_ANY> (jobs_dev, rep, grid, lds, s);
    else if ((rep.tile_variant & 3u) == smolhip::OUT_U4)
        launch_up_k<INK, smolhip::OUT_U4> (jobs_dev, rep, grid, lds, s);
    else
        launch_up_k<INK, smolhip::OUT_P4> (jobs_dev, rep, grid, lds, s);
#endif
}

} /* namespace */

namespace smolhip {
namespace launch {

/* grid.x counts tiles; a workgroup runs lind_waves () of them (one per wave) */
void
down (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    const uint32_t wv = smolhip::dispatch::lind_waves (rep);
    grid.x = (grid.x + wv - 1) / wv;
    if (rep.p16)
        launch_lind<2> (jobs_dev, rep, grid, lds, s);
    else if (rep.bpp_in == 3)
        launch_lind<1> (jobs_dev, rep, grid, lds, s);
    else
        launch_lind<0> (jobs_dev, rep, grid, lds, s);
}

void
up (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    if (rep.up_wide >= 2)
    {
        grid.x = (grid.x + 3) / 4;
        if (rep.up_wide == 3)
        {
            if (jobs_dev)
                hipLaunchKernelGGL (smolhip::k_upg_batch<true>, grid, dim3 (256), lds, s, jobs_dev);
            else
                hipLaunchKernelGGL (smolhip::k_upg_one<true>, grid, dim3 (256), lds, s, rep);
        }
        else
        {
            if (jobs_dev)
                hipLaunchKernelGGL (smolhip::k_upg_batch<false>, grid, dim3 (256), lds, s, jobs_dev);
            else
                hipLaunchKernelGGL (smolhip::k_upg_one<false>, grid, dim3 (256), lds, s, rep);
        }
        return;
    }
    if (rep.tile_aligned4)
        launch_up_in<smolhip::UP_IN_A4> (jobs_dev, rep, grid, lds, s);
    else
        launch_up_in<smolhip::UP_IN_ANY> (jobs_dev, rep, grid, lds, s);
}

} /* namespace launch */
} /* namespace smolhip */
