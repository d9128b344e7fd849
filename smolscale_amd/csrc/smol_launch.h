/* smol_launch.h -- launch wrappers of the kernel families, one translation unit each
 * (smol_k_stream.hip, smol_k_updown.hip, smol_k_tile0-3.hip, smol_k_box2.hip, smol_k_misc.hip). A launch
 * either reads descriptors from device memory (jobs_dev != NULL, one per blockIdx.y)
 * or takes the single descriptor `rep' by value as kernel argument. */

#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smol_jobdev.h"

namespace smolhip {
namespace launch {

/* grid = (column tiles of 64 thread columns, row tiles of block_rows output rows, jobs) */
void pow2 (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t block_rows, hipStream_t s);
void pow2x (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, hipStream_t s);
void up (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s);
void down (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s);
void tile_ts0 (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s);
void tile_ts1 (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s);
void tile_ts2 (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s);
void tile_ts3 (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s);
void tile_ts4 (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s);

/* k_tile: tapsel of tile_variant picks the translation unit */
inline void
tile (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s)
{
    switch (rep.tile_variant >> 5)
    {
        case 0:  tile_ts0 (jobs_dev, rep, grid, lds, s); break;
        case 1:  tile_ts1 (jobs_dev, rep, grid, lds, s); break;
        case 2:  tile_ts2 (jobs_dev, rep, grid, lds, s); break;
        case 3:  tile_ts3 (jobs_dev, rep, grid, lds, s); break;
        default: tile_ts4 (jobs_dev, rep, grid, lds, s); break;
    }
}
/* two passes through rep.tmp / jobs [i].tmp; h_tiles, v_tiles = waves of each pass (group maxima) */
void box2 (const JobDev *jobs_dev, const JobDev &rep, uint32_t n_jobs, uint32_t h_grid, uint32_t v_tiles,
           uint32_t h_lds, hipStream_t s);
void hspan (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s);
void wtile (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, uint32_t lds, hipStream_t s);
void wavepx (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, bool wide, hipStream_t s);
void general (const JobDev *jobs_dev, const JobDev &rep, dim3 grid, bool wide, hipStream_t s);
/* smol_hip_memory_zones' probe: `bytes_each' (a multiple of 16 KiB) read from a and from b side by side (smol_k_probe.hip) */
void zone_read2 (const void *a, const void *b, uint64_t bytes_each, hipStream_t s);

} /* namespace launch */
} /* namespace smolhip */
