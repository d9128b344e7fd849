// vspanbw.hip -- ceiling of "k_hspan with a vertical filter": a wave owns 256 columns of ONE output row and
// loads, itself, every source row that row needs (S = 2..11 rows x 256*rx pixels, coalesced 16-byte nt loads,
// all rows of a group in flight together), bounces each through its private LDS image, reads its taps
// (lane = columns l, l+64, l+128, l+192 so that tap addresses of neighbouring lanes are rx pixels apart),
// sums in registers and stores four dwords. No state carried between output rows, so rows shared by
// neighbouring output rows are fetched twice (from L2 / Infinity Cache, not HBM). Raster-order grid.
// Geometries: ratio 1.5, 3 (bilinear), 9.14 (box). Build: hipcc -O3 --offload-arch=gfx950 vspanbw.hip -o vspanbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
#define G1 __attribute__ ((address_space (1)))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e)); return 1; } } while (0)
extern __shared__ __attribute__ ((aligned (16))) uint8_t lds [];
struct Geo { const uint8_t *src; uint8_t *dst; uint32_t w_in, h_in, w_out, h_out; float rx, ry; uint32_t n_img, rows_per_out; };

// NW waves per workgroup (side by side on one output row), NL loads per lane and source row, NS pixel pairs per column,
// GR source rows loaded together
// SWZ (round 3): 1-D launch in the XCD-aware order -- workgroup o runs tile (o % 8) * (n / 8) + o / 8, so that each XCD takes a
// contiguous eighth of the (x, y, image) tiles and the source rows two neighbouring output rows share meet in ONE L2
template <int NW, int NL, int NS, int GR, bool SWZ = false>
__global__ void __launch_bounds__ (64 * NW) k_vspan (const Geo g, const uint32_t gx, const uint32_t gy, const uint32_t n)
{
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane (threadIdx.x >> 6);
    uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (SWZ)
    {
        const uint32_t o = blockIdx.x, q = n / 8, r = n % 8, xcd = o & 7;
        const uint32_t w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (o >> 3);
        bz = w / (gx * gy); const uint32_t rem = w - bz * gx * gy; by = rem / gx; bx = rem - by * gx;
    }
    const uint32_t x0 = (bx * NW + wave) * 256, y = by, img = bz;
    if (x0 >= g.w_out) return;
    const uint32_t S = g.rows_per_out;
    uint32_t r0 = (uint32_t) (y * g.ry); if (r0 + S > g.h_in) r0 = g.h_in - S;
    const uint32_t sx0 = (uint32_t) (x0 * g.rx);
    uint32_t span_px = (uint32_t) (256 * g.rx) + 2; if (sx0 + span_px > g.w_in) span_px = g.w_in - sx0;
    const uint32_t span_bytes = span_px * 4;
    uint8_t *img_lds = lds + wave * (GR * (NL * 1024 + 32));
    const G1 uint8_t *src = (const G1 uint8_t *) (uintptr_t) g.src + (size_t) img * g.w_in * g.h_in * 4;
    uint32_t tofs [4];
#pragma unroll
    for (int c = 0; c < 4; c++) { uint32_t x = x0 + c * 64 + lane; if (x >= g.w_out) x = g.w_out - 1; tofs [c] = ((uint32_t) (x * g.rx) - sx0) * 4; }
    uint32_t acc [4] = { 0, 0, 0, 0 };
    for (uint32_t s0 = 0; s0 < S; s0 += GR)
    {
        u32x4 v [GR][NL];
        uint32_t mis [GR];
#pragma unroll
        for (int r = 0; r < GR; r++)
        {
            const uint32_t row = (s0 + r < S) ? r0 + s0 + r : r0 + S - 1;
            const uintptr_t ra = (uintptr_t) (src + ((size_t) row * g.w_in + sx0) * 4);
            const uintptr_t a0 = ra & ~(uintptr_t) 15;
            mis [r] = (uint32_t) ra & 15u;
            const uint32_t nch = (mis [r] + span_bytes + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NL; i++)
                if (i * 64 + lane < nch) v [r][i] = __builtin_nontemporal_load ((const G1 u32x4 *) (a0 + (size_t) (i * 64 + lane) * 16));
        }
#pragma unroll
        for (int r = 0; r < GR; r++)
        {
            uint8_t *im = img_lds + r * (NL * 1024 + 32);
            const uint32_t nch = (mis [r] + span_bytes + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NL; i++)
                if (i * 64 + lane < nch) *reinterpret_cast<u32x4 *> (im + (size_t) (i * 64 + lane) * 16) = v [r][i];
        }
        asm volatile ("" ::: "memory");
#pragma unroll
        for (int r = 0; r < GR; r++)
        {
            const uint8_t *im = img_lds + r * (NL * 1024 + 32) + mis [r];
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int k = 0; k < NS; k++)
                {
                    const uint32_t *pp = reinterpret_cast<const uint32_t *> (im + tofs [c] + k * 8);
                    acc [c] += pp [0] ^ (pp [1] >> 1);
                }
        }
        asm volatile ("" ::: "memory");
    }
    G1 uint32_t *dp = (G1 uint32_t *) ((G1 uint8_t *) (uintptr_t) g.dst + (size_t) img * g.w_out * g.h_out * 4 + (size_t) y * g.w_out * 4);
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (x0 + c * 64 + lane < g.w_out) __builtin_nontemporal_store (acc [c], dp + x0 + c * 64 + lane);
}

static hipEvent_t e0, e1;
template <int NW, int NL, int NS, int GR, bool SWZ = false>
static void run (const char *name, Geo g)
{
    dim3 grid ((g.w_out + 256 * NW - 1) / (256 * NW), g.h_out, g.n_img);
    const uint32_t gx = grid.x, gy = grid.y, n = grid.x * grid.y * grid.z;
    if (SWZ) grid = dim3 (n);
    const size_t lds_b = (size_t) NW * GR * (NL * 1024 + 32);
    if (lds_b > 160 * 1024) { printf ("%-40s needs %zu KB LDS: skipped\n", name, lds_b / 1024); return; }
    (void) hipFuncSetAttribute ((const void *) k_vspan<NW, NL, NS, GR, SWZ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_b);
    auto launch = [&] { hipLaunchKernelGGL ((k_vspan<NW, NL, NS, GR, SWZ>), grid, dim3 (64 * NW), lds_b, 0, g, gx, gy, n); };
    for (int i = 0; i < 3; i++) launch ();
    (void) hipEventRecord (e0, 0);
    const int it = 20;
    for (int i = 0; i < it; i++) launch ();
    (void) hipEventRecord (e1, 0); (void) hipEventSynchronize (e1);
    float ms; (void) hipEventElapsedTime (&ms, e0, e1);
    const double us = ms * 1e3 / it, bytes = (double) g.n_img * 4.0 * ((double) g.w_in * g.h_in + (double) g.w_out * g.h_out);
    printf ("%-44s LDS %3zu KB  %8.1f us  %7.1f GB/s (%.1f %% of 8 TB/s)  %s\n", name, lds_b / 1024, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0, hipGetErrorString (hipGetLastError ()));
    fflush (stdout);
}

int main ()
{
    CK (hipEventCreate (&e0)); CK (hipEventCreate (&e1));
    const uint32_t n = 8;
    uint8_t *s, *d; CK (hipMalloc (&s, (size_t) n * 4096 * 4096 * 4 + 64)); CK (hipMalloc (&d, (size_t) n * 2731 * 2731 * 4 + 64));
    CK (hipMemset (s, 0x5a, (size_t) n * 4096 * 4096 * 4));
    Geo g3 = { s, d, 4096, 4096, 1365, 1365, 4096.f / 1365, 4096.f / 1365, n, 4 };
    run<1, 4, 2, 4> ("ratio 3   NW1 GR4 (4 rows per output row)", g3);
    run<2, 4, 2, 4> ("ratio 3   NW2 GR4", g3);
    run<4, 4, 2, 4> ("ratio 3   NW4 GR4", g3);
    run<4, 4, 2, 2> ("ratio 3   NW4 GR2", g3);
    run<6, 4, 2, 4> ("ratio 3   NW6 GR4 (whole row)", g3);
    run<1, 4, 2, 4, true> ("ratio 3   NW1 GR4 XCD-aware order", g3);
    run<2, 4, 2, 4, true> ("ratio 3   NW2 GR4 XCD-aware order", g3);
    run<4, 4, 2, 2, true> ("ratio 3   NW4 GR2 XCD-aware order", g3);
    Geo g15 = { s, d, 4096, 4096, 2731, 2731, 4096.f / 2731, 4096.f / 2731, 6, 2 };
    run<1, 2, 1, 2> ("ratio 1.5 NW1 GR2 (2 rows per output row)", g15);
    run<4, 2, 1, 2> ("ratio 1.5 NW4 GR2", g15);
    run<11, 2, 1, 2> ("ratio 1.5 NW11 GR2 (whole row)", g15);
    run<1, 2, 1, 2, true> ("ratio 1.5 NW1 GR2 XCD-aware order", g15);
    run<4, 2, 1, 2, true> ("ratio 1.5 NW4 GR2 XCD-aware order", g15);
    Geo g9 = { s, d, 4096, 4096, 448, 448, 4096.f / 448, 4096.f / 448, n, 11 };
    run<1, 10, 4, 2> ("ratio 9.14 NW1 GR2 (11 rows per output row)", g9);
    run<2, 10, 4, 2> ("ratio 9.14 NW2 GR2", g9);
    run<2, 10, 4, 3> ("ratio 9.14 NW2 GR3", g9);
    run<2, 10, 4, 1> ("ratio 9.14 NW2 GR1", g9);
    run<1, 10, 4, 2, true> ("ratio 9.14 NW1 GR2 XCD-aware order", g9);
    run<2, 10, 4, 2, true> ("ratio 9.14 NW2 GR2 XCD-aware order", g9);
    run<2, 10, 4, 4, true> ("ratio 9.14 NW2 GR4 XCD-aware order", g9);
    Geo g6 = { s, d, 4096, 4096, 683, 683, 4096.f / 683, 4096.f / 683, n, 7 };
    run<1, 7, 4, 2> ("ratio 6    NW1 GR2 (7 rows per output row)", g6);
    run<3, 7, 4, 2> ("ratio 6    NW3 GR2 (whole row)", g6);
    run<3, 7, 4, 4> ("ratio 6    NW3 GR4", g6);
    run<1, 7, 4, 2, true> ("ratio 6    NW1 GR2 XCD-aware order", g6);
    run<3, 7, 4, 2, true> ("ratio 6    NW3 GR2 XCD-aware order", g6);
    run<1, 7, 4, 7, true> ("ratio 6    NW1 GR7 (all rows at once) XCD-aware order", g6);
    return 0;
}
