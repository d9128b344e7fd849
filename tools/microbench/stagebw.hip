// stagebw.hip -- how fast can a one-wave workgroup pull row segments into LDS? (design input for the tiled kernels)
// Pattern of BASELINE config 4: image rows of 196605 bytes, a tile = 16 rows x 1536 bytes, 32768 tiles.
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), ring of RING rows, wait per row, nothing else
//   mode 1: global_load_dwordx4 into registers, ds_write_b128, ring of RING rows in registers
//   mode 2: global_load_dwordx4 into registers only (xor-reduce, no LDS)
// Build: hipcc -O3 --offload-arch=gfx950 stagebw.hip -o stagebw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e)); return 1; } } while (0)
constexpr uint32_t STRIDE = 196605, ROWS = 4096, TILE_ROWS = 16, SEG = 1536, TILES_X = 128;
extern __shared__ __attribute__ ((aligned (16))) uint8_t lds [];

__device__ __forceinline__ void dma16 (const void *g, uint8_t *l)
{
    __builtin_amdgcn_global_load_lds (reinterpret_cast<const __attribute__ ((address_space (1))) void *> (reinterpret_cast<uintptr_t> (g)),
                                      (__attribute__ ((address_space (3))) void *) l, 16, 0, 0);
}

template <int RING, int STORE = 0>
__global__ void __launch_bounds__ (64) k_dma (const uint8_t *src, uint32_t *sink, uint32_t *dst = nullptr)
{
    const uint32_t lane = threadIdx.x, tx = blockIdx.x % TILES_X, ty = blockIdx.x / TILES_X;
    uintptr_t ra = (uintptr_t) src + (size_t) ty * TILE_ROWS * STRIDE + (size_t) tx * SEG;
    uint32_t acc = 0;
    uint32_t outv [TILE_ROWS];
    auto issue = [&] (uint32_t slot) {
        const uintptr_t a0 = ra & ~(uintptr_t) 15;
        const uint32_t nch = (((uint32_t) ra & 15u) + SEG + 15u) >> 4;      // 96 or 97 chunks
        uint8_t *dst = lds + slot * 1568;
        dma16 ((const void *) (a0 + lane * 16), dst);
        if (64 + lane < nch) dma16 ((const void *) (a0 + (64 + lane) * 16), dst + 1024);
        ra += STRIDE;
    };
#pragma unroll
    for (int i = 0; i < RING - 1; i++) issue (i);
#pragma unroll
    for (uint32_t r = 0; r < TILE_ROWS; r++)
    {
        // rows retire in order: allow the (RING-2) younger rows (2 instructions each) to stay in flight
        constexpr bool ST = false && (STORE == 1 || STORE == 2);
        if (RING == 2) { if (ST) asm volatile ("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile ("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (RING == 3) { if (ST) asm volatile ("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile ("s_waitcnt vmcnt(2)" ::: "memory"); }
        if (RING == 4) { if (ST) asm volatile ("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile ("s_waitcnt vmcnt(4)" ::: "memory"); }
        if (RING == 6) asm volatile ("s_waitcnt vmcnt(8)" ::: "memory");
        if (RING == 8) asm volatile ("s_waitcnt vmcnt(12)" ::: "memory");
        if (r + RING - 1 < TILE_ROWS) issue ((r + RING - 1) % RING);
        else { dma16 ((const void *) ((uintptr_t) src + lane * 16), lds + RING * 1568); dma16 ((const void *) ((uintptr_t) src + lane * 16), lds + RING * 1568); } // keep the count uniform
        const uint32_t v = *reinterpret_cast<const uint32_t *> (lds + (r % RING) * 1568 + lane * 24);
        acc ^= v;
        if (STORE == 1 || STORE == 2)
        {
            // one 256-byte output row per source row (config 4: 8192 x 4096 x 4 bytes out)
            uint32_t *o = dst + ((size_t) (ty * TILE_ROWS + r) * TILES_X + tx) * 64 + lane;
            if (STORE == 2) __builtin_nontemporal_store (v, o); else *o = v;
        }
        if (STORE == 3) outv [r] = v;
        if (STORE == 6)
        {
            // as STORE 1, output rows padded by 256 bytes (row stride 33024 instead of 32768): power-of-two stride test
            dst [(size_t) (ty * TILE_ROWS + r) * (TILES_X * 64 + 64) + tx * 64 + lane] = v;
        }
        if (STORE == 4 && (tx & 3u) == 0)
        {
            // same bytes as STORE 1, but in 1-KB runs: every fourth tile writes 16 bytes per lane
            u32x4 *o = reinterpret_cast<u32x4 *> (dst) + ((size_t) (ty * TILE_ROWS + r) * (TILES_X / 4) + tx / 4) * 64 + lane;
            *o = u32x4 { v, v, v, v };
        }
        if (STORE == 5 && (tx & 15u) == 0)
        {
            // ... in 4-KB runs: every sixteenth tile writes four 16-byte vectors per lane
            u32x4 *o = reinterpret_cast<u32x4 *> (dst) + ((size_t) (ty * TILE_ROWS + r) * (TILES_X / 16) + tx / 16) * 256 + lane;
            o [0] = u32x4 { v, v, v, v }; o [64] = u32x4 { v, v, v, v }; o [128] = u32x4 { v, v, v, v }; o [192] = u32x4 { v, v, v, v };
        }
    }
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
    if (STORE == 3)
    {
#pragma unroll
        for (uint32_t r = 0; r < TILE_ROWS; r++)
            dst [((size_t) (ty * TILE_ROWS + r) * TILES_X + tx) * 64 + lane] = outv [r];
    }
    if (acc == 0x12345679u) *sink = acc;
}

// the same traffic with tiles 4x wider: 6 KB of every row per wave (7 DMA instructions), 1-KB stores
template <int RING, int ROWS_T>
__global__ void __launch_bounds__ (64) k_dma_wide (const uint8_t *src, uint32_t *sink, uint32_t *dst)
{
    constexpr uint32_t SEGW = SEG * 4, PITCH = SEGW + 32, TXW = TILES_X / 4;
    const uint32_t lane = threadIdx.x, tx = blockIdx.x % TXW, ty = blockIdx.x / TXW;
    uintptr_t ra = (uintptr_t) src + (size_t) ty * ROWS_T * STRIDE + (size_t) tx * SEGW;
    uint32_t acc = 0;
    auto issue = [&] (uint32_t slot) {
        const uintptr_t a0 = ra & ~(uintptr_t) 15;
        const uint32_t nch = (((uint32_t) ra & 15u) + SEGW + 15u) >> 4;      // 384 or 385 chunks
        uint8_t *d = lds + slot * PITCH;
#pragma unroll
        for (uint32_t k = 0; k < 6; k++) dma16 ((const void *) (a0 + (k * 64 + lane) * 16), d + k * 1024);
        if (384 + lane < nch) dma16 ((const void *) (a0 + (384 + lane) * 16), d + 6144);
        else if (lane == 0) dma16 ((const void *) (a0), d + 6144 + 16);     // keep the count uniform (7 per row)
        ra += STRIDE;
    };
#pragma unroll
    for (int i = 0; i < RING - 1; i++) issue (i);
#pragma unroll 1
    for (uint32_t r = 0; r < ROWS_T; r++)
    {
        if (RING == 2) asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
        if (RING == 3) asm volatile ("s_waitcnt vmcnt(7)" ::: "memory");
        if (r + RING - 1 < ROWS_T) issue ((r + RING - 1) % RING);
        const u32x4 v = *reinterpret_cast<const u32x4 *> (lds + (r % RING) * PITCH + lane * 96);
        acc ^= v.x;
        reinterpret_cast<u32x4 *> (dst) [((size_t) (ty * ROWS_T + r) * TXW + tx) * 64 + lane] = v;
    }
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345679u) *sink = acc;
}

template <int DEPTH, bool TO_LDS, bool STORE = false>
__global__ void __launch_bounds__ (64) k_reg (const uint8_t *src, uint32_t *sink, uint32_t *dst = nullptr)
{
    const uint32_t lane = threadIdx.x, tx = blockIdx.x % TILES_X, ty = blockIdx.x / TILES_X;
    const uintptr_t base = ((uintptr_t) src + (size_t) ty * TILE_ROWS * STRIDE + (size_t) tx * SEG);
    u32x4 buf [DEPTH][2];
    uint32_t acc = 0;
    auto load = [&] (uint32_t r, u32x4 (&b) [2]) {
        const uintptr_t ra = base + (size_t) r * STRIDE;
        const uintptr_t a0 = ra & ~(uintptr_t) 15;
        b [0] = __builtin_nontemporal_load ((const __attribute__ ((address_space (1))) u32x4 *) (a0 + lane * 16));
        b [1] = (lane < 33) ? __builtin_nontemporal_load ((const __attribute__ ((address_space (1))) u32x4 *) (a0 + (64 + lane) * 16)) : u32x4 { 0, 0, 0, 0 };
    };
#pragma unroll
    for (int i = 0; i < DEPTH; i++) load (i, buf [i]);
#pragma unroll
    for (uint32_t r = 0; r < TILE_ROWS; r++)
    {
        u32x4 (&b) [2] = buf [r % DEPTH];
        if (TO_LDS)
        {
            *reinterpret_cast<u32x4 *> (lds + (r & 1) * 1568 + lane * 16) = b [0];
            if (lane < 33) *reinterpret_cast<u32x4 *> (lds + (r & 1) * 1568 + 1024 + lane * 16) = b [1];
            const uint32_t v = *reinterpret_cast<const uint32_t *> (lds + (r & 1) * 1568 + lane * 24);
            acc ^= v;
            if (STORE) __builtin_nontemporal_store (v, dst + ((size_t) (ty * TILE_ROWS + r) * TILES_X + tx) * 64 + lane);
        }
        else
            acc ^= b [0].x ^ b [0].w ^ b [1].y;
        if (r + DEPTH < TILE_ROWS) load (r + DEPTH, b);
    }
    if (acc == 0x12345679u) *sink = acc;
}

int main ()
{
    const size_t bytes = (size_t) STRIDE * ROWS + 4096;
    uint8_t *src; uint32_t *sink;
    CK (hipMalloc (&src, bytes)); CK (hipMalloc (&sink, 4)); CK (hipMemset (src, 1, bytes));
    hipEvent_t e0, e1; CK (hipEventCreate (&e0)); CK (hipEventCreate (&e1));
    const unsigned grid = TILES_X * (ROWS / TILE_ROWS);
    const double traffic = (double) SEG * TILES_X * ROWS;
    auto run = [&] (const char *name, auto launch) {
        for (int i = 0; i < 50; i++) launch ();
        hipEventRecord (e0, 0);
        const int it = 50;
        for (int i = 0; i < it; i++) launch ();
        hipEventRecord (e1, 0); hipEventSynchronize (e1);
        float ms = 0; (void) hipEventElapsedTime (&ms, e0, e1);
        printf ("%-44s %8.1f us  %7.1f GB/s\n", name, ms * 1e3 / it, traffic / (ms * 1e-3 / it) / 1e9);
    };
    run ("LDS-DMA ring 2 (1 row in flight)", [&] { hipLaunchKernelGGL (k_dma<2>, dim3 (grid), dim3 (64), 3 * 1568, 0, src, sink); });
    run ("LDS-DMA ring 3", [&] { hipLaunchKernelGGL (k_dma<3>, dim3 (grid), dim3 (64), 4 * 1568, 0, src, sink); });
    run ("LDS-DMA ring 4", [&] { hipLaunchKernelGGL (k_dma<4>, dim3 (grid), dim3 (64), 5 * 1568, 0, src, sink); });
    run ("LDS-DMA ring 6", [&] { hipLaunchKernelGGL (k_dma<6>, dim3 (grid), dim3 (64), 7 * 1568, 0, src, sink); });
    run ("LDS-DMA ring 8", [&] { hipLaunchKernelGGL (k_dma<8>, dim3 (grid), dim3 (64), 9 * 1568, 0, src, sink); });
    uint32_t *dst; CK (hipMalloc (&dst, (size_t) ROWS * (TILES_X * 256 + 256) + 4096));
    run ("LDS-DMA ring 2 + plain 256-B store per row", [&] { hipLaunchKernelGGL ((k_dma<2, 1>), dim3 (grid), dim3 (64), 3 * 1568, 0, src, sink, dst); });
    run ("LDS-DMA ring 4 + plain 256-B store per row", [&] { hipLaunchKernelGGL ((k_dma<4, 1>), dim3 (grid), dim3 (64), 5 * 1568, 0, src, sink, dst); });
    run ("LDS-DMA ring 2 + nt 256-B store per row", [&] { hipLaunchKernelGGL ((k_dma<2, 2>), dim3 (grid), dim3 (64), 3 * 1568, 0, src, sink, dst); });
    run ("LDS-DMA ring 4 + nt 256-B store per row", [&] { hipLaunchKernelGGL ((k_dma<4, 2>), dim3 (grid), dim3 (64), 5 * 1568, 0, src, sink, dst); });
    run ("LDS-DMA ring 4 + same bytes in 1-KB runs (1/4 of the waves)", [&] { hipLaunchKernelGGL ((k_dma<4, 4>), dim3 (grid), dim3 (64), 5 * 1568, 0, src, sink, dst); });
    run ("LDS-DMA ring 4 + same bytes in 4-KB runs (1/16 of the waves)", [&] { hipLaunchKernelGGL ((k_dma<4, 5>), dim3 (grid), dim3 (64), 5 * 1568, 0, src, sink, dst); });
    run ("LDS-DMA ring 4 + plain 256-B store, output stride 32768+256", [&] { hipLaunchKernelGGL ((k_dma<4, 6>), dim3 (grid), dim3 (64), 5 * 1568, 0, src, sink, dst); });
    run ("256-column tiles x 16 rows, ring 2, 1-KB stores", [&] { hipLaunchKernelGGL ((k_dma_wide<2, 16>), dim3 (grid / 4), dim3 (64), 2 * 6176 + 64, 0, src, sink, dst); });
    run ("256-column tiles x 4 rows, ring 2, 1-KB stores", [&] { hipLaunchKernelGGL ((k_dma_wide<2, 4>), dim3 (grid), dim3 (64), 2 * 6176 + 64, 0, src, sink, dst); });
    run ("LDS-DMA ring 2 + 16 stores at the tile's end", [&] { hipLaunchKernelGGL ((k_dma<2, 3>), dim3 (grid), dim3 (64), 3 * 1568, 0, src, sink, dst); });
    run ("LDS-DMA ring 4 + 16 stores at the tile's end", [&] { hipLaunchKernelGGL ((k_dma<4, 3>), dim3 (grid), dim3 (64), 5 * 1568, 0, src, sink, dst); });
    run ("registers depth 2 -> ds_write", [&] { hipLaunchKernelGGL ((k_reg<2, true>), dim3 (grid), dim3 (64), 2 * 1568, 0, src, sink); });
    run ("registers depth 4 -> ds_write", [&] { hipLaunchKernelGGL ((k_reg<4, true>), dim3 (grid), dim3 (64), 2 * 1568, 0, src, sink); });
    run ("registers depth 8 -> ds_write", [&] { hipLaunchKernelGGL ((k_reg<8, true>), dim3 (grid), dim3 (64), 2 * 1568, 0, src, sink); });
    run ("registers depth 2 -> ds_write + nt 256-B store per row", [&] { hipLaunchKernelGGL ((k_reg<2, true, true>), dim3 (grid), dim3 (64), 2 * 1568, 0, src, sink, dst); });
    run ("registers depth 4 -> ds_write + nt 256-B store per row", [&] { hipLaunchKernelGGL ((k_reg<4, true, true>), dim3 (grid), dim3 (64), 2 * 1568, 0, src, sink, dst); });
    run ("registers depth 8 -> ds_write + nt 256-B store per row", [&] { hipLaunchKernelGGL ((k_reg<8, true, true>), dim3 (grid), dim3 (64), 2 * 1568, 0, src, sink, dst); });
    run ("registers depth 4, no LDS", [&] { hipLaunchKernelGGL ((k_reg<4, false>), dim3 (grid), dim3 (64), 0, 0, src, sink); });
    run ("registers depth 8, no LDS", [&] { hipLaunchKernelGGL ((k_reg<8, false>), dim3 (grid), dim3 (64), 0, 0, src, sink); });
    run ("registers depth 16 (whole tile), no LDS", [&] { hipLaunchKernelGGL ((k_reg<16, false>), dim3 (grid), dim3 (64), 0, 0, src, sink); });
    return 0;
}
