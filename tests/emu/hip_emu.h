/* hip_emu.h -- TEST INFRASTRUCTURE: just enough of the HIP device vocabulary to
 * compile smolscale_amd/csrc/smol_kernels.hip.h as ordinary host C++ (clang++),
 * so kernel logic can be stepped through without a GPU. Blocks and threads are
 * executed one after the other by emu_kernels.cpp. */
#pragma once
#include <stdint.h>

#define __device__
#define __global__
#define __forceinline__ inline
#define __constant__ static const
#define __launch_bounds__(...)
#define __shared__ static

struct emu_dim3 { unsigned x, y, z; };
static thread_local emu_dim3 blockIdx, threadIdx, blockDim, gridDim;

struct uint2 { uint32_t x, y; };
static inline uint2 make_uint2 (uint32_t x, uint32_t y) { uint2 r = { x, y }; return r; }

/* v_perm_b32: result byte i = byte sel[i] of the 8-byte value {S0:S1} (S1 = low dword) */
static inline uint32_t __builtin_amdgcn_perm (uint32_t s0, uint32_t s1, uint32_t sel)
{
    uint64_t both = ((uint64_t) s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++)
    {
        uint32_t k = (sel >> (8 * i)) & 0xff;
        uint32_t b = k < 8 ? (uint32_t) (both >> (8 * k)) & 0xff : (k >= 0x0d ? 0xff : 0);
        r |= b << (8 * i);
    }
    return r;
}
