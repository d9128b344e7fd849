#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// does ds_read_b64 / ds_read_b32 / ds_read_b96 return the right bytes at byte-granular addresses on gfx950, and at what cost?
__global__ void k (uint32_t *out, uint64_t *cyc)
{
    __shared__ __attribute__ ((aligned (16))) uint8_t lds [8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds [i] = (uint8_t) (i * 7 + 3);
    __syncthreads ();
    const uint32_t lane = threadIdx.x;
    const uint32_t base = (uint32_t) (uintptr_t) (__attribute__ ((address_space (3))) uint8_t *) lds;
    uint32_t bad = 0;
    for (uint32_t off = 0; off < 16; off++)
    {
        const uint32_t a = lane * 24 + off;      // per-lane byte address, any alignment
        uint64_t v; uint32_t w;
        asm volatile ("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v" (v) : "v" (base + a) : "memory");
        asm volatile ("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v" (w) : "v" (base + a) : "memory");
        uint64_t want = 0; uint32_t want32 = 0;
        for (int b = 0; b < 8; b++) want |= (uint64_t) (uint8_t) ((a + b) * 7 + 3) << (8 * b);
        for (int b = 0; b < 4; b++) want32 |= (uint32_t) (uint8_t) ((a + b) * 7 + 3) << (8 * b);
        if (v != want) bad |= 1u << off;
        if (w != want32) bad |= 1u << (16 + off);
    }
    out [lane] = bad;
    // timing: 256 reads at aligned vs misaligned addresses
    for (int mis = 0; mis < 4; mis++)
    {
        uint32_t acc = 0;
        uint64_t t0 = __builtin_readcyclecounter ();
        for (int it = 0; it < 256; it++)
        {
            uint64_t v;
            const uint32_t a = ((lane * 24 + it * 8) & 4095) + mis;
            asm volatile ("ds_read_b64 %0, %1" : "=v" (v) : "v" (a) : "memory");
            asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc ^= (uint32_t) v;
        }
        uint64_t t1 = __builtin_readcyclecounter ();
        if (lane == 0) cyc [mis] = t1 - t0;
        out [64 + lane] ^= acc;
    }
}
int main ()
{
    uint32_t *out; uint64_t *cyc; hipMalloc (&out, 1024); hipMalloc (&cyc, 64);
    hipLaunchKernelGGL (k, dim3 (1), dim3 (64), 0, 0, out, cyc);
    uint32_t h [64]; uint64_t c [4];
    hipMemcpy (h, out, 256, hipMemcpyDeviceToHost); hipMemcpy (c, cyc, 32, hipMemcpyDeviceToHost);
    uint32_t any = 0; for (int i = 0; i < 64; i++) any |= h [i];
    printf ("lane0 mask %08x lane1 %08x\n", h [0], h [1]);
    printf ("mismatch mask (bit off = b64 at offset off wrong, bit 16+off = b32 wrong): %08x\n", any);
    printf ("ticks for 256 dependent ds_read_b64 at byte offset 0/1/2/3: %llu %llu %llu %llu\n", (unsigned long long) c [0], (unsigned long long) c [1], (unsigned long long) c [2], (unsigned long long) c [3]);
    return 0;
}
