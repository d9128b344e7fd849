// spanbw.hip -- the row-span structure for vertical-COPY jobs (BASELINE config 4): a wave owns 256
// adjacent output pixels of ONE row = 6144 contiguous source bytes (RGB8, 8 source pixels per output
// pixel), loads them with 6-7 coalesced 16-byte loads per lane, all in flight, passes them through a
// private LDS image (each lane then owns a 96-byte window = its 4 output pixels), and stores 16 bytes
// per lane = 1 KiB contiguous per wave. Upper bound of what such a kernel can reach, with the real
// read and write volumes and no arithmetic.
//   ALIGN 0: loads start at the byte-exact first source byte (unaligned 16-byte loads)
//   ALIGN 1: loads start at the 16-byte boundary below it (one more load per lane)
//   LDS 0/1: without / with the LDS round trip (ds_write_b128 x N, ds_read_b128 x 6)
//   SPANS  : consecutive 256-pixel spans one wave walks along its row (loads of span s+1 issued before span s is consumed)
// Build: hipcc -O3 --offload-arch=gfx950 spanbw.hip -o spanbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
#define G1 __attribute__ ((address_space (1)))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e)); return 1; } } while (0)
constexpr uint32_t W_IN = 65535, H = 4096, W_OUT = 8192, STRIDE = W_IN * 3, OSTRIDE = W_OUT * 4, SEG = 6144;
extern __shared__ __attribute__ ((aligned (16))) uint8_t lds [];

template <int ALIGN, int LDS, int SPANS, int WAVES, int NT>
__global__ void __launch_bounds__ (64 * WAVES) k_span (const uint8_t *src, uint8_t *dst)
{
    constexpr int NL = ALIGN ? 7 : 6;
    typedef u32x4 UV __attribute__ ((aligned (1)));
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr uint32_t spans_per_row = W_OUT / 256;                  // 32
    constexpr uint32_t groups_per_row = spans_per_row / SPANS;
    const uint32_t gid = blockIdx.x * WAVES + wave;                  // one wave = SPANS consecutive spans of one row
    const uint32_t y = gid / groups_per_row, g = gid % groups_per_row;
    if (y >= H) return;
    uint8_t *my = lds + wave * (7 * 1024 + 16);
    const G1 uint8_t *row = (const G1 uint8_t *) (uintptr_t) src + (size_t) y * STRIDE;
    G1 uint8_t *orow = (G1 uint8_t *) (uintptr_t) dst + (size_t) y * OSTRIDE;
    u32x4 v [2][NL];
    auto issue = [&] (int slot, uint32_t span) {
        uintptr_t a = (uintptr_t) (row + (size_t) span * SEG);
        if (ALIGN) a &= ~(uintptr_t) 15;
#pragma unroll
        for (int i = 0; i < NL; i++)
        {
            const G1 uint8_t *p = (const G1 uint8_t *) (a + (size_t) (i * 64 + lane) * 16);
            if (ALIGN) v [slot][i] = NT ? __builtin_nontemporal_load ((const G1 u32x4 *) p) : *(const G1 u32x4 *) p;
            else       v [slot][i] = NT ? __builtin_nontemporal_load ((const G1 UV *) p) : *(const G1 UV *) p;
        }
    };
    issue (0, g * SPANS);
#pragma unroll
    for (int s = 0; s < SPANS; s++)
    {
        if (s + 1 < SPANS) issue ((s + 1) & 1, g * SPANS + s + 1);
        u32x4 o = { 0, 0, 0, 0 };
        if (LDS)
        {
#pragma unroll
            for (int i = 0; i < NL; i++)
                *reinterpret_cast<u32x4 *> (my + (i * 64 + lane) * 16) = v [s & 1][i];
            __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
            const uint32_t mis = ALIGN ? ((uint32_t) (uintptr_t) (row + (size_t) (g * SPANS + s) * SEG) & 15u) & ~15u : 0u;   // (timing only)
#pragma unroll
            for (int i = 0; i < 6; i++)
            {
                const u32x4 w = *reinterpret_cast<const u32x4 *> (my + lane * 96 + i * 16 + mis);
                o.x ^= w.x + i; o.y += w.y; o.z ^= w.z; o.w += w.w ^ w.x;
            }
            __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
        }
        else
        {
#pragma unroll
            for (int i = 0; i < NL; i++) { o.x ^= v [s & 1][i].x; o.y += v [s & 1][i].y; o.z ^= v [s & 1][i].z; o.w += v [s & 1][i].w; }
        }
        G1 u32x4 *op = (G1 u32x4 *) (orow + (size_t) (g * SPANS + s) * 1024) + lane;
        if (NT) __builtin_nontemporal_store (o, op); else *op = o;
    }
}

static hipEvent_t e0, e1;
template <int ALIGN, int LDS, int SPANS, int WAVES, int NT>
static void run (const char *name, const uint8_t *s, uint8_t *d)
{
    const uint32_t waves = H * (W_OUT / 256 / SPANS);
    const uint32_t grid = (waves + WAVES - 1) / WAVES;
    const size_t lds_b = LDS ? WAVES * (7 * 1024 + 16) : 0;
    auto launch = [&] { hipLaunchKernelGGL ((k_span<ALIGN, LDS, SPANS, WAVES, NT>), dim3 (grid), dim3 (64 * WAVES), lds_b, 0, s, d); };
    for (int i = 0; i < 3; i++) launch ();
    (void) hipEventRecord (e0, 0);
    const int it = 20;
    for (int i = 0; i < it; i++) launch ();
    (void) hipEventRecord (e1, 0); (void) hipEventSynchronize (e1);
    float ms; (void) hipEventElapsedTime (&ms, e0, e1);
    const double us = ms * 1e3 / it, bytes = (double) W_IN * 3 * H + (double) W_OUT * 4 * H;
    printf ("%-56s %8.1f us  %7.1f GB/s (%.1f %% of 8 TB/s)\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0);
    fflush (stdout);
}

int main ()
{
    CK (hipEventCreate (&e0)); CK (hipEventCreate (&e1));
    const size_t in_b = (size_t) STRIDE * H + 4096, out_b = (size_t) OSTRIDE * H;
    uint8_t *s, *d; CK (hipMalloc (&s, in_b)); CK (hipMalloc (&d, out_b)); CK (hipMemset (s, 0x5a, in_b));
    run<1, 0, 1, 1, 1> ("aligned   noLDS spans1 W1 nt", s, d);
    run<1, 0, 1, 4, 1> ("aligned   noLDS spans1 W4 nt", s, d);
    run<1, 0, 2, 4, 1> ("aligned   noLDS spans2 W4 nt", s, d);
    run<1, 0, 4, 4, 1> ("aligned   noLDS spans4 W4 nt", s, d);
    run<1, 0, 4, 4, 0> ("aligned   noLDS spans4 W4 plain", s, d);
    run<0, 0, 1, 1, 1> ("unaligned noLDS spans1 W1 nt", s, d);
    run<0, 0, 1, 4, 1> ("unaligned noLDS spans1 W4 nt", s, d);
    run<0, 0, 4, 4, 1> ("unaligned noLDS spans4 W4 nt", s, d);
    run<0, 0, 4, 4, 0> ("unaligned noLDS spans4 W4 plain", s, d);
    run<1, 1, 1, 1, 1> ("aligned   LDS   spans1 W1 nt", s, d);
    run<1, 1, 1, 4, 1> ("aligned   LDS   spans1 W4 nt", s, d);
    run<1, 1, 2, 4, 1> ("aligned   LDS   spans2 W4 nt", s, d);
    run<1, 1, 4, 4, 1> ("aligned   LDS   spans4 W4 nt", s, d);
    run<1, 1, 4, 2, 1> ("aligned   LDS   spans4 W2 nt", s, d);
    run<1, 1, 8, 4, 1> ("aligned   LDS   spans8 W4 nt", s, d);
    run<0, 1, 1, 4, 1> ("unaligned LDS   spans1 W4 nt", s, d);
    run<0, 1, 4, 4, 1> ("unaligned LDS   spans4 W4 nt", s, d);
    run<0, 1, 8, 4, 1> ("unaligned LDS   spans8 W4 nt", s, d);
    return 0;
}
