// gatherbw.hip -- which MEMORY STRUCTURE can feed a bilinear-minification kernel at the copy rate?
// Candidates for the rebuild of k_tile, C4 geometry (65535x4096 RGB8, odd stride ->
// 8192x4096 RGBA8, 4 samples of 2 pixels per output pixel) and a 32-bit ratio-3 geometry:
//   gather   lane = output column; per row NS independent 8-byte loads at the byte-exact address of
//            each sample's pixel pair (scalar row base + per-lane constant offset), RB rows in flight;
//            no LDS. Relies on unaligned global loads.
//   gather3  same, from the 4-byte aligned-down address, 12 bytes per sample (what a kernel must do if
//            unaligned loads are slow).
//   window   lane loads its contiguous NS*2*BPP-byte window with 16/8-byte unaligned loads.
// Every variant reduces what it loaded to one dword per lane and row and stores it (non-temporal):
// the floor of the access pattern with the real read and write volumes. Build:
//   hipcc -O3 --offload-arch=gfx950 gatherbw.hip -o gatherbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
typedef uint32_t u32x2 __attribute__ ((ext_vector_type (2)));
typedef uint32_t u32x3 __attribute__ ((ext_vector_type (3)));
#define G1 __attribute__ ((address_space (1)))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e)); return 1; } } while (0)

struct Geo { const uint8_t *src; uint8_t *dst; uint32_t w_in, h, w_out, bpp, stride_in, stride_out, band; uint64_t step; };

template <typename T> __device__ __forceinline__ T ld_unaligned (const G1 uint8_t *p)
{
    typedef T UT __attribute__ ((aligned (1)));
    return __builtin_nontemporal_load (reinterpret_cast<const G1 UT *> (p));
}

// MODE 0 gather (8 B byte-exact), 1 gather3 (12 B from aligned-down), 2 window (contiguous), 3 gather 8 B aligned-down (wrong bytes, timing only)
template <int MODE, int NS, int RB, int WAVES>
__global__ void __launch_bounds__ (64 * WAVES) k_gather (const Geo g)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tiles_x = (g.w_out + 64 * WAVES - 1) / (64 * WAVES);
    const uint32_t tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    uint32_t x = (tx * WAVES + wave) * 64 + lane;
    const bool live = x < g.w_out;
    if (!live) x = g.w_out - 1;
    uint32_t off [NS];
#pragma unroll
    for (int s = 0; s < NS; s++)
    {
        uint64_t pos = (uint64_t) (x * NS + s) * g.step;
        uint32_t o = (uint32_t) (pos >> 32);
        if (o > g.w_in - 2) o = g.w_in - 2;
        off [s] = o * g.bpp;
    }
    const uint32_t y0 = ty * g.band, y1 = min (y0 + g.band, g.h);
    const G1 uint8_t *rowp = (const G1 uint8_t *) (uintptr_t) g.src + (size_t) y0 * g.stride_in;
    G1 uint8_t *outp = (G1 uint8_t *) (uintptr_t) g.dst + (size_t) y0 * g.stride_out + (size_t) x * 4;
    for (uint32_t y = y0; y < y1; y += RB)
    {
        uint32_t acc [RB];
        if (MODE == 0 || MODE == 3)
        {
            u32x2 v [RB][NS];
#pragma unroll
            for (int r = 0; r < RB; r++)
#pragma unroll
                for (int s = 0; s < NS; s++)
                {
                    const G1 uint8_t *p = rowp + (size_t) r * g.stride_in + off [s];
                    if (MODE == 3) p = (const G1 uint8_t *) ((uintptr_t) p & ~(uintptr_t) 3);
                    v [r][s] = ld_unaligned<u32x2> (p);
                }
#pragma unroll
            for (int r = 0; r < RB; r++)
            {
                acc [r] = 0;
#pragma unroll
                for (int s = 0; s < NS; s++) acc [r] += v [r][s].x ^ (v [r][s].y >> 3);
            }
        }
        else if (MODE == 1)
        {
            u32x3 v [RB][NS];
#pragma unroll
            for (int r = 0; r < RB; r++)
#pragma unroll
                for (int s = 0; s < NS; s++)
                {
                    const uintptr_t a = (uintptr_t) (rowp + (size_t) r * g.stride_in + off [s]);
                    v [r][s] = __builtin_nontemporal_load (reinterpret_cast<const G1 u32x3 *> (a & ~(uintptr_t) 3));
                }
#pragma unroll
            for (int r = 0; r < RB; r++)
            {
                acc [r] = 0;
#pragma unroll
                for (int s = 0; s < NS; s++) acc [r] += v [r][s].x ^ (v [r][s].y >> 3) ^ (v [r][s].z << 5);
            }
        }
        else
        {
            // window of NS * 2 * bpp bytes starting at off [0]: 16-byte pieces (last one 8 bytes when that covers it)
            constexpr int NB = NS * 2 * 4;                   // upper bound (bpp 4); for bpp 3 NS*6 <= NB
            constexpr int N16 = (NB + 15) / 16;
            u32x4 v [RB][N16];
#pragma unroll
            for (int r = 0; r < RB; r++)
#pragma unroll
                for (int k = 0; k < N16; k++)
                    if (k * 16 < (int) (NS * 2 * g.bpp))
                        v [r][k] = ld_unaligned<u32x4> (rowp + (size_t) r * g.stride_in + off [0] + k * 16);
#pragma unroll
            for (int r = 0; r < RB; r++)
            {
                acc [r] = 0;
#pragma unroll
                for (int k = 0; k < N16; k++)
                    if (k * 16 < (int) (NS * 2 * g.bpp))
                        acc [r] += v [r][k].x ^ (v [r][k].y >> 3) ^ v [r][k].z ^ (v [r][k].w << 2);
            }
        }
#pragma unroll
        for (int r = 0; r < RB; r++)
            if (live && y + r < y1)
                __builtin_nontemporal_store (acc [r], reinterpret_cast<G1 uint32_t *> (outp + (size_t) r * g.stride_out));
        rowp += (size_t) RB * g.stride_in;
        outp += (size_t) RB * g.stride_out;
    }
}

static hipEvent_t e0, e1;
template <typename F> static float time_us (F launch)
{
    for (int i = 0; i < 3; i++) launch ();
    hipEventRecord (e0, 0);
    const int it = 20;
    for (int i = 0; i < it; i++) launch ();
    hipEventRecord (e1, 0); hipEventSynchronize (e1);
    float ms; hipEventElapsedTime (&ms, e0, e1);
    return ms * 1e3f / it;
}

template <int MODE, int NS, int RB, int WAVES>
static void run (const char *name, Geo g, uint32_t band, int n_img, size_t img_in, size_t img_out)
{
    g.band = band;
    const uint32_t tiles_x = (g.w_out + 64 * WAVES - 1) / (64 * WAVES);
    const uint32_t grid = tiles_x * ((g.h + band - 1) / band);
    const float us = time_us ([&] {
        for (int i = 0; i < n_img; i++)
        {
            Geo gi = g; gi.src = g.src + i * img_in; gi.dst = g.dst + i * img_out;
            hipLaunchKernelGGL ((k_gather<MODE, NS, RB, WAVES>), dim3 (grid), dim3 (64 * WAVES), 0, 0, gi);
        }
    });
    const double bytes = (double) n_img * ((double) g.w_in * g.h * g.bpp + (double) g.w_out * g.h * 4);
    printf ("%-44s band %4u  %8.1f us  %7.1f GB/s (%.1f %% of 8 TB/s)\n", name, band, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0);
    fflush (stdout);
}

int main (int argc, char **argv)
{
    CK (hipEventCreate (&e0)); CK (hipEventCreate (&e1));
    const char *which = argc > 1 ? argv [1] : "c4";
    if (!strcmp (which, "c4"))
    {
        Geo g = {};
        g.w_in = 65535; g.h = 4096; g.w_out = 8192; g.bpp = 3; g.stride_in = 65535 * 3; g.stride_out = 8192 * 4;
        g.step = ((uint64_t) g.w_in << 32) / (g.w_out * 4);
        const size_t in_b = (size_t) g.stride_in * g.h + 64, out_b = (size_t) g.stride_out * g.h;
        uint8_t *s, *d; CK (hipMalloc (&s, in_b)); CK (hipMalloc (&d, out_b)); CK (hipMemset (s, 0x5a, in_b));
        g.src = s; g.dst = d;
        for (uint32_t band : { 16u, 64u, 256u })
        {
            run<0, 4, 1, 1> ("c4 gather 8B exact   NS4 RB1 W1", g, band, 1, 0, 0);
            run<0, 4, 2, 1> ("c4 gather 8B exact   NS4 RB2 W1", g, band, 1, 0, 0);
            run<0, 4, 4, 1> ("c4 gather 8B exact   NS4 RB4 W1", g, band, 1, 0, 0);
            run<0, 4, 2, 4> ("c4 gather 8B exact   NS4 RB2 W4", g, band, 1, 0, 0);
            run<0, 4, 4, 4> ("c4 gather 8B exact   NS4 RB4 W4", g, band, 1, 0, 0);
            run<3, 4, 4, 1> ("c4 gather 8B aligned NS4 RB4 W1", g, band, 1, 0, 0);
            run<1, 4, 2, 1> ("c4 gather 12B down   NS4 RB2 W1", g, band, 1, 0, 0);
            run<1, 4, 4, 1> ("c4 gather 12B down   NS4 RB4 W1", g, band, 1, 0, 0);
            run<2, 4, 2, 1> ("c4 window 24B        NS4 RB2 W1", g, band, 1, 0, 0);
            run<2, 4, 4, 1> ("c4 window 24B        NS4 RB4 W1", g, band, 1, 0, 0);
            run<2, 4, 8, 1> ("c4 window 24B        NS4 RB8 W1", g, band, 1, 0, 0);
            run<2, 4, 4, 4> ("c4 window 24B        NS4 RB4 W4", g, band, 1, 0, 0);
        }
    }
    else
    {
        // 32-bit pixels, ratio 3 horizontally (2 samples per column), every source row feeds the pass: 8 images
        Geo g = {};
        g.w_in = 4096; g.h = 4096; g.w_out = 1365; g.bpp = 4; g.stride_in = 4096 * 4; g.stride_out = 1365 * 4;
        g.step = ((uint64_t) g.w_in << 32) / (g.w_out * 2);
        const int n = 8;
        const size_t in_b = (size_t) g.stride_in * g.h, out_b = (size_t) g.stride_out * g.h;
        uint8_t *s, *d; CK (hipMalloc (&s, in_b * n + 64)); CK (hipMalloc (&d, out_b * n)); CK (hipMemset (s, 0x5a, in_b * n));
        g.src = s; g.dst = d;
        for (uint32_t band : { 16u, 64u, 256u })
        {
            run<0, 2, 2, 1> ("r3 gather 8B NS2 RB2 W1", g, band, n, in_b, out_b);
            run<0, 2, 4, 1> ("r3 gather 8B NS2 RB4 W1", g, band, n, in_b, out_b);
            run<0, 2, 8, 1> ("r3 gather 8B NS2 RB8 W1", g, band, n, in_b, out_b);
            run<0, 2, 4, 4> ("r3 gather 8B NS2 RB4 W4", g, band, n, in_b, out_b);
            run<0, 2, 8, 4> ("r3 gather 8B NS2 RB8 W4", g, band, n, in_b, out_b);
        }
    }
    return 0;
}
