// rowsbw.hip -- ceiling of the "rows" structure for arbitrary-ratio minification (successor of k_tile):
// a workgroup of NW waves owns a strip of 256 output columns x R output rows; its S = R*ry + 2 source
// row segments (256*rx pixels each) are loaded at once, one row per wave in turn (coalesced 16-byte nt
// loads -> private LDS image -> "taps": each lane reads 4 x NS pixel pairs at its columns' offsets and
// writes 4 packed pixels to the shared H-result array), barrier, then each wave emits output rows
// (reads two H rows, stores 16 bytes per lane). No arithmetic beyond xor: the floor of the access
// pattern, with the real LDS traffic. Geometries: ratio 3 (4096^2 -> 1365^2 x 8) and 9.14 (4096^2 -> 448^2 x 8).
// Build: hipcc -O3 --offload-arch=gfx950 rowsbw.hip -o rowsbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
#define G1 __attribute__ ((address_space (1)))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e)); return 1; } } while (0)
extern __shared__ __attribute__ ((aligned (16))) uint8_t lds [];

struct Geo { const uint8_t *src; uint8_t *dst; uint32_t w_in, h_in, w_out, h_out; float rx, ry; uint32_t n_img; };

template <int NW, int R, int NL, int NS>
__global__ void __launch_bounds__ (64 * NW) k_rows (const Geo g)
{
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane (threadIdx.x >> 6);
    const uint32_t tiles_x = (g.w_out + 255) / 256;
    const uint32_t tx = blockIdx.x, ty = blockIdx.y, img = blockIdx.z;
    const uint32_t y0 = ty * R;
    if (y0 >= g.h_out) return;
    const uint32_t rows_out = min ((uint32_t) R, g.h_out - y0);
    const uint32_t r0 = (uint32_t) (y0 * g.ry);
    uint32_t r1 = (uint32_t) ((y0 + rows_out) * g.ry) + 1; if (r1 > g.h_in - 1) r1 = g.h_in - 1;
    const uint32_t S = r1 - r0 + 1;
    const uint32_t x0 = tx * 256, sx0 = (uint32_t) (x0 * g.rx);
    uint32_t span_px = (uint32_t) (256 * g.rx) + 2; if (sx0 + span_px > g.w_in) span_px = g.w_in - sx0;
    const uint32_t span_bytes = span_px * 4;
    uint8_t *img_lds = lds + wave * (NL * 1024 + 32);
    uint32_t *hres = reinterpret_cast<uint32_t *> (lds + NW * (NL * 1024 + 32));      // [S][256]
    const G1 uint8_t *src = (const G1 uint8_t *) (uintptr_t) g.src + (size_t) img * g.w_in * g.h_in * 4;
    // per-lane tap byte offsets for 4 column groups
    uint32_t tofs [4];
#pragma unroll
    for (int c = 0; c < 4; c++) tofs [c] = ((uint32_t) ((x0 + c * 64 + lane) * g.rx) - sx0) * 4;
    for (uint32_t s = wave; s < S; s += NW)
    {
        const uintptr_t ra = (uintptr_t) (src + ((size_t) (r0 + s) * g.w_in + sx0) * 4);
        const uintptr_t a0 = ra & ~(uintptr_t) 15;
        const uint32_t mis = (uint32_t) ra & 15u, nch = (mis + span_bytes + 15) >> 4;
        u32x4 v [NL];
#pragma unroll
        for (int i = 0; i < NL; i++)
            if (i * 64 + lane < nch) v [i] = __builtin_nontemporal_load ((const G1 u32x4 *) (a0 + (size_t) (i * 64 + lane) * 16));
#pragma unroll
        for (int i = 0; i < NL; i++)
            if (i * 64 + lane < nch) *reinterpret_cast<u32x4 *> (img_lds + (size_t) (i * 64 + lane) * 16) = v [i];
        asm volatile ("" ::: "memory");
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < NS; k++)
            {
                const uint32_t *pp = reinterpret_cast<const uint32_t *> (img_lds + mis + tofs [c] + k * 8);
                acc += pp [0] ^ (pp [1] >> 1);
            }
            hres [s * 256 + c * 64 + lane] = acc;
        }
        asm volatile ("" ::: "memory");
    }
    __syncthreads ();
    for (uint32_t r = wave; r < rows_out; r += NW)
    {
        const uint32_t sa = min ((uint32_t) (r * g.ry), S - 2);
        const u32x4 a = *reinterpret_cast<const u32x4 *> (hres + sa * 256 + lane * 4);
        const u32x4 b = *reinterpret_cast<const u32x4 *> (hres + (sa + 1) * 256 + lane * 4);
        const u32x4 o = { a.x ^ b.x, a.y + b.y, a.z ^ b.z, a.w + b.w };
        const uint32_t x = x0 + lane * 4;
        G1 uint8_t *dp = (G1 uint8_t *) (uintptr_t) g.dst + (size_t) img * g.w_out * g.h_out * 4 + ((size_t) (y0 + r) * g.w_out + x) * 4;
        if (x + 4 <= g.w_out) { typedef u32x4 UV __attribute__ ((aligned (4))); __builtin_nontemporal_store (o, (G1 UV *) dp); }
        else for (uint32_t k = 0; x + k < g.w_out; k++) ((G1 uint32_t *) dp) [k] = o [k];
    }
    (void) tiles_x;
}

static hipEvent_t e0, e1;
template <int NW, int R, int NL, int NS>
static void run (const char *name, Geo g)
{
    const dim3 grid ((g.w_out + 255) / 256, (g.h_out + R - 1) / R, g.n_img);
    const uint32_t S_max = (uint32_t) (R * g.ry) + 4;
    const size_t lds_b = NW * (NL * 1024 + 32) + (size_t) S_max * 1024;
    if (lds_b > 160 * 1024) { printf ("%-40s needs %zu KB LDS: skipped\n", name, lds_b / 1024); return; }
    (void) hipFuncSetAttribute ((const void *) k_rows<NW, R, NL, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_b);
    auto launch = [&] { hipLaunchKernelGGL ((k_rows<NW, R, NL, NS>), grid, dim3 (64 * NW), lds_b, 0, g); };
    for (int i = 0; i < 3; i++) launch ();
    (void) hipEventRecord (e0, 0);
    const int it = 20;
    for (int i = 0; i < it; i++) launch ();
    (void) hipEventRecord (e1, 0); (void) hipEventSynchronize (e1);
    float ms; (void) hipEventElapsedTime (&ms, e0, e1);
    const double us = ms * 1e3 / it, bytes = (double) g.n_img * 4.0 * ((double) g.w_in * g.h_in + (double) g.w_out * g.h_out);
    printf ("%-40s LDS %3zu KB  %8.1f us  %7.1f GB/s (%.1f %% of 8 TB/s)  %s\n", name, lds_b / 1024, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0, hipGetErrorString (hipGetLastError ()));
    fflush (stdout);
}

int main ()
{
    CK (hipEventCreate (&e0)); CK (hipEventCreate (&e1));
    const uint32_t n = 8;
    uint8_t *s, *d; CK (hipMalloc (&s, (size_t) n * 4096 * 4096 * 4 + 64)); CK (hipMalloc (&d, (size_t) n * 2731 * 2731 * 4 + 64));
    CK (hipMemset (s, 0x5a, (size_t) n * 4096 * 4096 * 4));
    Geo g3 = { s, d, 4096, 4096, 1365, 1365, 4096.f / 1365, 4096.f / 1365, n };
    run<8, 8, 4, 2> ("ratio 3   NW8 R8", g3);
    run<8, 4, 4, 2> ("ratio 3   NW8 R4", g3);
    run<8, 2, 4, 2> ("ratio 3   NW8 R2", g3);
    run<4, 4, 4, 2> ("ratio 3   NW4 R4", g3);
    run<4, 2, 4, 2> ("ratio 3   NW4 R2", g3);
    run<16, 8, 4, 2> ("ratio 3   NW16 R8", g3);
    run<16, 4, 4, 2> ("ratio 3   NW16 R4", g3);
    Geo g15 = { s, d, 4096, 4096, 2731, 2731, 4096.f / 2731, 4096.f / 2731, 6 };
    run<8, 8, 2, 1> ("ratio 1.5 NW8 R8", g15);
    run<8, 4, 2, 1> ("ratio 1.5 NW8 R4", g15);
    run<4, 4, 2, 1> ("ratio 1.5 NW4 R4", g15);
    run<16, 8, 2, 1> ("ratio 1.5 NW16 R8", g15);
    Geo g9 = { s, d, 4096, 4096, 448, 448, 4096.f / 448, 4096.f / 448, n };
    run<8, 2, 10, 4> ("ratio 9.14 NW8 R2", g9);
    run<8, 1, 10, 4> ("ratio 9.14 NW8 R1", g9);
    run<4, 1, 10, 4> ("ratio 9.14 NW4 R1", g9);
    run<16, 2, 10, 4> ("ratio 9.14 NW16 R2", g9);
    run<16, 4, 10, 4> ("ratio 9.14 NW16 R4", g9);
    return 0;
}
