/* smol_k_tile.inc.h -- kernel instantiations + launch wrappers of the k_tile (arbitrary ratios, box, copy; 64bpp class) family (gfx950).
 * One translation unit per kernel family so that `make -j' compiles them side by
 * side; smol_hip.hip (the runtime) only sees smol_launch.h. */

#include <hip/hip_runtime.h>

#include "smol_kernels.hip.h"
#include "smol_kernel_tile.hip.h"
#include "smol_launch.h"

using smolhip::JobDev;

namespace {

/* k_tile has one entry point per (tapsel, horizontal mode, columns per lane) */
template <int TS, int HM, int CPL>
void
launch_tile_k (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    if (jobs_dev)
        hipLaunchKernelGGL ((smolhip::k_tile_batch<TS, HM, CPL>), grid, dim3 (64), lds, s, jobs_dev);
    else
        hipLaunchKernelGGL ((smolhip::k_tile_one<TS, HM, CPL>), grid, dim3 (64), lds, s, rep);
}

/* ... and per compiled-in vertical mode (smol_jobdev.h: TILE_VM_*) */
template <int TS, int HM, int CPL, int VM>
void
launch_tile_kv (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    if (jobs_dev)
        hipLaunchKernelGGL ((smolhip::k_tile_batch<TS, HM, CPL, VM>), grid, dim3 (64), lds, s, jobs_dev);
    else
        hipLaunchKernelGGL ((smolhip::k_tile_one<TS, HM, CPL, VM>), grid, dim3 (64), lds, s, rep);
}

template <int TS>
void
launch_tile_ts (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    const uint32_t hm = (rep.tile_variant >> 2) & 7u;

    using namespace smolhip;

    if (rep.tile_cpl == 4 && rep.tile_vm == TILE_VM_PAIR)
    {
        switch (hm)
        {
            case 1:  launch_tile_kv<TS, 1, 4, TILE_VM_PAIR> (jobs_dev, rep, grid, lds, s); break;
            case 2:  launch_tile_kv<TS, 2, 4, TILE_VM_PAIR> (jobs_dev, rep, grid, lds, s); break;
            default: launch_tile_kv<TS, 3, 4, TILE_VM_PAIR> (jobs_dev, rep, grid, lds, s); break;
        }
        return;
    }
    if (rep.tile_cpl == 4)
    {
        switch (hm)
        {
            case 0:  launch_tile_k<TS, 0, 4> (jobs_dev, rep, grid, lds, s); break;
            case 1:  launch_tile_k<TS, 1, 4> (jobs_dev, rep, grid, lds, s); break;
            case 2:  launch_tile_k<TS, 2, 4> (jobs_dev, rep, grid, lds, s); break;
            default: launch_tile_k<TS, 3, 4> (jobs_dev, rep, grid, lds, s); break;
        }
        return;
    }
    if (rep.tile_vm == TILE_VM_COPY)
    {
        /* horizontal bilinear minification with a vertical copy (width-only resizes k_hspan does not take) */
        switch (hm)
        {
            case 1:  launch_tile_kv<TS, 1, 1, TILE_VM_COPY> (jobs_dev, rep, grid, lds, s); break;
            case 2:  launch_tile_kv<TS, 2, 1, TILE_VM_COPY> (jobs_dev, rep, grid, lds, s); break;
            default: launch_tile_kv<TS, 3, 1, TILE_VM_COPY> (jobs_dev, rep, grid, lds, s); break;
        }
        return;
    }
    if (rep.tile_vm == TILE_VM_PAIR)
    {
        /* bilinear on both axes: the arbitrary-ratio minification everybody runs */
        switch (hm)
        {
            case 1:  launch_tile_kv<TS, 1, 1, TILE_VM_PAIR> (jobs_dev, rep, grid, lds, s); break;
            case 2:  launch_tile_kv<TS, 2, 1, TILE_VM_PAIR> (jobs_dev, rep, grid, lds, s); break;
            default: launch_tile_kv<TS, 3, 1, TILE_VM_PAIR> (jobs_dev, rep, grid, lds, s); break;
        }
        return;
    }
    if (rep.tile_vm == TILE_VM_BOX)
    {
        if (hm == 4)
            launch_tile_kv<TS, 4, 1, TILE_VM_BOX> (jobs_dev, rep, grid, lds, s);
        else
            launch_tile_kv<TS, 5, 1, TILE_VM_BOX> (jobs_dev, rep, grid, lds, s);
        return;
    }
    switch (hm)
    {
        case 0:  launch_tile_k<TS, 0, 1> (jobs_dev, rep, grid, lds, s); break;
        case 1:  launch_tile_k<TS, 1, 1> (jobs_dev, rep, grid, lds, s); break;
        case 2:  launch_tile_k<TS, 2, 1> (jobs_dev, rep, grid, lds, s); break;
        case 3:  launch_tile_k<TS, 3, 1> (jobs_dev, rep, grid, lds, s); break;
        case 4:  launch_tile_k<TS, 4, 1> (jobs_dev, rep, grid, lds, s); break;
        default: launch_tile_k<TS, 5, 1> (jobs_dev, rep, grid, lds, s); break;
    }
}

} /* namespace */

namespace smolhip {
namespace launch {

/* one translation unit per tap reader (SMOL_TILE_TS = 0 A4, 1 A4 + premultiply while
 * staging, 2 U4, 3 U3, 4 A4 after an in-LDS premultiply pass): smol_k_tile0.hip ... smol_k_tile4.hip */
#define SMOL_TILE_FN2(n) tile_ts##n
#define SMOL_TILE_FN(n) SMOL_TILE_FN2 (n)

void
SMOL_TILE_FN (SMOL_TILE_TS) (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    launch_tile_ts<SMOL_TILE_TS> (jobs_dev, rep, grid, lds, s);
}

} /* namespace launch */
} /* namespace smolhip */
