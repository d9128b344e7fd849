/* smol_k_tile.inc.h -- kernel instantiations + launch wrappers of the k_tile (arbitrary ratios, box, copy; 64bpp class) family (gfx950).
 * One translation unit per kernel family so that `make -j' compiles them side by
 * side; the runtime (smol_rt_*.hip) only sees smol_launch.h. */

#include <hip/hip_runtime.h>

#include "smol_kernels.hip.h"
#include "smol_kernel_tile.hip.h"
#include "smol_launch.h"

using smolhip::JobDev;

namespace {

/* k_tile has one entry point per (tapsel, horizontal mode, columns per lane) */
/* Workgroup shape. The entry points of the plain 32-bit, one-sample, one-column, bilinear-vertical family
 * (ratios 1-2: the lightest per-tile work) run 4 (or 3, 2) waves per workgroup on neighbouring column tiles, each in
 * its own LDS region -- no workgroup barriers in these variants -- when the column tiles divide into such groups
 * with at most 6 % of the wave slots left empty: neighbouring tiles then share a CU and its L2 instead of landing on
 * different XCDs (4096^2 -> 2409^2 109 -> 94 us, -> 3413^2 147 -> 131, -> 3034^2 124 -> 116). Everything else keeps
 * one wave per workgroup: measured mixed elsewhere, from +7 % (ratios 3.5, 6, 7) to -16 % (photographs, rgbA
 * sources, ratio 5: profiles/r02z_tile_wg.txt). */
template <int TS, int HM, int CPL, int VM>
inline uint32_t
tile_wg_shape (dim3 &grid, dim3 &block, uint32_t &lds_total, uint32_t lds_per_wave)
{
    uint32_t nw = 1;
    const bool family = smolhip::tile_multiwave_variant (TS == 0 ? 0 : 1, HM, TS == 0 ? 0 : 1, CPL, VM);
    if (family && !smolhip::tune_env ("SMOL_TILE_WG1"))
    {
        for (uint32_t w = 4; w >= 2 && nw == 1; w--)
        {
            const uint32_t slots = (grid.x + w - 1) / w * w;
            if (w * lds_per_wave <= 64u * 1024u && (slots - grid.x) * 100u <= slots * 6u)
                nw = w;
        }
    }
    block = dim3 (64 * nw);
    grid.x = (grid.x + nw - 1) / nw;
    lds_total = nw * lds_per_wave;
    return nw;
}

/* XCD-aware order (smol_jobdev.h: TileSwz): a 1-D launch of all workgroups, each XCD taking a contiguous eighth */
inline smolhip::TileSwz
tile_swizzle (dim3 &grid)
{
    uint32_t n;
    const smolhip::TileSwz z = smolhip::swizzle_make (grid.x, grid.y, grid.z, !smolhip::tune_env ("SMOL_TILE_NO_SWZ"), &n);
    if (n)
        grid = dim3 (n);
    return z;
}

template <int TS, int HM, int CPL>
void
launch_tile_k (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    dim3 block;
    uint32_t lds_total;
    const uint32_t nw = tile_wg_shape<TS, HM, CPL, smolhip::TILE_VM_ANY> (grid, block, lds_total, lds);
    const smolhip::TileSwz swz = tile_swizzle (grid);
    if (jobs_dev)
        hipLaunchKernelGGL ((smolhip::k_tile_batch<TS, HM, CPL>), grid, block, lds_total, s, jobs_dev, lds, nw, swz);
    else
        hipLaunchKernelGGL ((smolhip::k_tile_one<TS, HM, CPL>), grid, block, lds_total, s, rep, lds, nw, swz);
}

/* ... and per compiled-in vertical mode (smol_jobdev.h: TILE_VM_*) */
template <int TS, int HM, int CPL, int VM>
void
launch_tile_kv (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    dim3 block;
    uint32_t lds_total;
    const uint32_t nw = tile_wg_shape<TS, HM, CPL, VM> (grid, block, lds_total, lds);
    const smolhip::TileSwz swz = tile_swizzle (grid);
    if (jobs_dev)
        hipLaunchKernelGGL ((smolhip::k_tile_batch<TS, HM, CPL, VM>), grid, block, lds_total, s, jobs_dev, lds, nw, swz);
    else
        hipLaunchKernelGGL ((smolhip::k_tile_one<TS, HM, CPL, VM>), grid, block, lds_total, s, rep, lds, nw, swz);
}

template <int TS>
void
launch_tile_ts (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    const uint32_t hm = (rep.tile_variant >> 2) & 7u;

    using namespace smolhip;

    if (rep.tile_cpl == 2)
    {
        /* two columns per lane (round 5): bilinear on both axes only (tile_setup) */
        switch (hm)
        {
            case 1:  launch_tile_kv<TS, 1, 2, TILE_VM_PAIR> (jobs_dev, rep, grid, lds, s); break;
            case 2:  launch_tile_kv<TS, 2, 2, TILE_VM_PAIR> (jobs_dev, rep, grid, lds, s); break;
            default: launch_tile_kv<TS, 3, 2, TILE_VM_PAIR> (jobs_dev, rep, grid, lds, s); break;
        }
        return;
    }
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
