// patternbw.hip -- the floor of a tiled resampler's ACCESS PATTERN: one-wave workgroups that only move the
// bytes a 64-column tile of k_tile moves (source row segments into LDS by LDS-DMA, double-buffered; one 256-byte
// store per output row), no arithmetic. Usage: patternbw <w_in> <h_in> <bpp_in> <stride_in> <w_out> <h_out> <rows_per_tile> <frames> [columns_per_tile = 64 | 128 | 256]
// Build: hipcc -O3 --offload-arch=gfx950 patternbw.hip -o patternbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e)); return 1; } } while (0)
extern __shared__ __attribute__ ((aligned (16))) uint8_t lds [];
struct P { const uint8_t *src; uint32_t *dst; uint32_t w_in, h_in, bpp, stride, w_out, h_out, tr, tiles_x, tiles_y, tc, band; size_t frame_in, frame_out; };

__device__ __forceinline__ void dma16 (const void *g, uint8_t *l)
{
    __builtin_amdgcn_global_load_lds (reinterpret_cast<const __attribute__ ((address_space (1))) void *> (reinterpret_cast<uintptr_t> (g)),
                                      (__attribute__ ((address_space (3))) void *) l, 16, 0, 0);
}

__global__ void __launch_bounds__ (64) k_pattern (const P p, uint32_t *sink)
{
    // band = 1: each XCD (blocks go round-robin over the 8 of them) takes a contiguous band of tiles
    const uint32_t bid = p.band ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t lane = threadIdx.x, tx = bid % p.tiles_x, ty = bid / p.tiles_x, f = blockIdx.y;
    if (ty >= p.tiles_y) return;
    const uint32_t y0 = ty * p.tr, y1 = min (y0 + p.tr, p.h_out);
    // source span of the tile: columns [64 tx, 64 tx + 64) and rows [y0, y1) scaled by the ratios
    const uint32_t TC = p.tc, cpl = TC / 64u;                  // columns per tile / per lane
    const uint32_t sx0 = (uint32_t) ((uint64_t) (TC * tx) * p.w_in / p.w_out);
    const uint32_t sx1 = min ((uint32_t) ((uint64_t) min (TC * tx + TC, p.w_out) * p.w_in / p.w_out) + 1u, p.w_in - 1u);
    const uint32_t r0 = (uint32_t) ((uint64_t) y0 * p.h_in / p.h_out), r1 = min ((uint32_t) ((uint64_t) y1 * p.h_in / p.h_out) + 1u, p.h_in - 1u);
    const uint32_t seg = (sx1 - sx0 + 1) * p.bpp, pitch = (seg + 47u) & ~15u;
    uintptr_t ra = (uintptr_t) p.src + f * p.frame_in + (size_t) r0 * p.stride + (size_t) sx0 * p.bpp;
    uint32_t acc = 0, out_r = y0;
    auto issue = [&] (uint32_t slot) {
        const uintptr_t a0 = ra & ~(uintptr_t) 15;
        const uint32_t nch = (((uint32_t) ra & 15u) + seg + 15u) >> 4;
        for (uint32_t k0 = 0; k0 < nch; k0 += 64) if (k0 + lane < nch) dma16 ((const void *) (a0 + (k0 + lane) * 16), lds + slot * pitch + k0 * 16);
        ra += p.stride;
    };
    issue (0);
    for (uint32_t r = r0; r <= r1; r++)
    {
        asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
        if (r < r1) issue ((r - r0 + 1) & 1u);
        const uint32_t v = *reinterpret_cast<const uint32_t *> (lds + ((r - r0) & 1u) * pitch + (lane * (seg / 64u) & ~3u));
        acc ^= v;
        // an output row leaves whenever the source rows it needs have passed
        while (out_r < y1 && (uint32_t) ((uint64_t) (out_r + 1) * p.h_in / p.h_out) <= r + 1)
        {
            for (uint32_t c = 0; c < cpl; c++)
                if (TC * tx + lane * cpl + c < p.w_out)
                    p.dst [f * (p.frame_out / 4) + (size_t) out_r * p.w_out + TC * tx + lane * cpl + c] = v;
            out_r++;
        }
    }
    if (acc == 0x12345679u) *sink = acc;
}

int main (int argc, char **argv)
{
    if (argc < 9) { printf ("usage: patternbw w_in h_in bpp_in stride_in w_out h_out rows_per_tile frames\n"); return 1; }
    P p; p.w_in = atoi (argv [1]); p.h_in = atoi (argv [2]); p.bpp = atoi (argv [3]); p.stride = atoi (argv [4]);
    p.w_out = atoi (argv [5]); p.h_out = atoi (argv [6]); p.tr = atoi (argv [7]); const uint32_t frames = atoi (argv [8]);
    p.tc = argc > 9 ? atoi (argv [9]) : 64;
    p.band = argc > 10 ? atoi (argv [10]) : 0;
    p.tiles_x = (p.w_out + p.tc - 1) / p.tc; p.tiles_y = (p.h_out + p.tr - 1) / p.tr;
    p.frame_in = ((size_t) p.stride * p.h_in + 4095) & ~(size_t) 4095; p.frame_out = (size_t) p.w_out * p.h_out * 4;
    uint8_t *src; uint32_t *dst, *sink;
    CK (hipMalloc (&src, p.frame_in * frames + 4096)); CK (hipMalloc (&dst, p.frame_out * frames + 4096)); CK (hipMalloc (&sink, 4));
    CK (hipMemset (src, 1, p.frame_in * frames)); p.src = src; p.dst = dst;
    const uint32_t span = (uint32_t) ((uint64_t) p.tc * p.w_in / p.w_out) + 3;
    const uint32_t lds_bytes = 2 * ((span * p.bpp + 47u) & ~15u) + 64;
    hipEvent_t e0, e1; CK (hipEventCreate (&e0)); CK (hipEventCreate (&e1));
    const dim3 grid ((p.tiles_x * p.tiles_y + 7u) & ~7u, frames);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL (k_pattern, grid, dim3 (64), lds_bytes, 0, p, sink);
    hipEventRecord (e0, 0);
    const int it = 50;
    for (int i = 0; i < it; i++) hipLaunchKernelGGL (k_pattern, grid, dim3 (64), lds_bytes, 0, p, sink);
    hipEventRecord (e1, 0); hipEventSynchronize (e1);
    float ms = 0; (void) hipEventElapsedTime (&ms, e0, e1);
    const double bytes = (double) frames * ((double) p.w_in * p.h_in * p.bpp + (double) p.w_out * p.h_out * 4);
    printf ("%s%ux%u (%u B/px, stride %u) -> %ux%u x%u, %u-column x %u-row tiles: %.1f us, %.0f GB/s algorithmic (%.1f %% of 8 TB/s)\n",
            p.band ? "[XCD bands] " : "", p.w_in, p.h_in, p.bpp, p.stride, p.w_out, p.h_out, frames, p.tc, p.tr, ms * 1e3 / it, bytes / (ms * 1e-3 / it) / 1e9, bytes / (ms * 1e-3 / it) / 8e12 * 100);
    return 0;
}
