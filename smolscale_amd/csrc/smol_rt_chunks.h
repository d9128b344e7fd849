/* smol_rt_chunks.h -- host-only planning helpers of the road from host memory (smol_rt_lanes.hip), kept apart from the
 * runtime so that tests/emu can replay them without a GPU. */

#ifndef SMOLSCALE_AMD_SMOL_RT_CHUNKS_H
#define SMOLSCALE_AMD_SMOL_RT_CHUNKS_H

#include <stddef.h>
#include <stdint.h>

#include "smol_hip.h"

namespace smolrt {

/* The mapped road's chunks (smol_rt_lanes.hip, strip_issue_locked): output rows per chunk such that what a chunk keeps in a ring
 * slot -- its source rows when the source is pageable, its output rows when the destination is -- fits `slot_max' bytes, starting from
 * about `want' bytes of traffic per chunk and halving. rows == 0: not even one output row fits (the caller takes the DMA road). */
struct MappedChunks { uint32_t rows; size_t in_max, out_max; };

inline MappedChunks
mapped_chunks (const SmolHipJob &job, uint32_t first_row, uint32_t n_rows, size_t in_row_bytes, size_t out_row_bytes,
               size_t src_pitch, size_t dst_pitch, bool ring_in, bool ring_out, size_t want, size_t slot_max)
{
    MappedChunks mc = { n_rows, 0, 0 };
    uint32_t r0 = 0, r1 = 0;

    if (!n_rows || smol_hip_input_row_range (&job, first_row, n_rows, &r0, &r1))
    {
        mc.rows = 0;
        return mc;
    }
    if (!ring_in && !ring_out)
        return mc;                      /* page-locked on both sides: one launch */
    const size_t per_row = (((size_t) r1 - r0 + 1) * in_row_bytes + (size_t) n_rows * out_row_bytes) / n_rows + 1;
    if (want > slot_max)
        want = slot_max;
    const uint64_t rows0 = want / per_row;
    mc.rows = rows0 < 1 ? 1u : rows0 > n_rows ? n_rows : (uint32_t) rows0;
    for (;;)
    {
        mc.in_max = mc.out_max = 0;
        for (uint32_t y = 0; y < n_rows; y += mc.rows)
        {
            const uint32_t rows_here = n_rows - y < mc.rows ? n_rows - y : mc.rows;
            uint32_t a = 0, b = 0;
            if (smol_hip_input_row_range (&job, first_row + y, rows_here, &a, &b))
            {
                mc.rows = 0;
                return mc;
            }
            const size_t in_b = ring_in ? ((size_t) b - a + 1) * src_pitch : 0;
            const size_t out_b = ring_out ? (size_t) rows_here * dst_pitch : 0;
            if (in_b > mc.in_max) mc.in_max = in_b;
            if (out_b > mc.out_max) mc.out_max = out_b;
        }
        if (mc.in_max <= slot_max && mc.out_max <= slot_max)
            return mc;
        if (mc.rows == 1)
        {
            mc.rows = 0;                /* (one output row reads more than a slot holds) */
            return mc;
        }
        mc.rows = (mc.rows + 1) / 2;
    }
}

} /* namespace smolrt */

#endif
