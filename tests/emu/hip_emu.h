/* hip_emu.h -- TEST INFRASTRUCTURE: just enough of the HIP device vocabulary to
 * compile smolscale_amd/csrc/smol_kernels.hip.h as ordinary host C++ (clang++),
 * so kernel logic can be stepped through without a GPU. Blocks and threads are
 * executed one after the other by emu_kernels.cpp. */
#pragma once
#include <stdint.h>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __constant__ const
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

/* v_alignbyte_b32: ({S0,S1} >> (8 * S2[1:0])) & 0xffffffff */
static inline uint32_t __builtin_amdgcn_alignbyte (uint32_t s0, uint32_t s1, uint32_t sh)
{
    uint64_t both = ((uint64_t) s0 << 32) | s1;
    return (uint32_t) (both >> (8 * (sh & 3)));
}

/* Block-level barrier + dynamic LDS for kernels whose lanes cooperate: the
 * harness runs the lanes of ONE block as real threads (emu_kernels.cpp). */
#include <pthread.h>
extern pthread_barrier_t emu_block_barrier;
extern uint8_t *emu_dyn_lds_ptr;
static inline void __syncthreads (void) { pthread_barrier_wait (&emu_block_barrier); }
#define smol_dyn_lds emu_dyn_lds_ptr

static inline uint32_t __umul24 (uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }

/* LDS-DMA stand-ins: synchronous copy of this lane's 16 bytes */
#include <string.h>
static inline void smol_global_load_lds16 (const void *gptr, uint8_t *lds_wave_base, uint32_t lane)
{
    memcpy (lds_wave_base + 16 * lane, gptr, 16);
}
static inline void smol_wait_vmcnt0 (void) {}
template <int N> static inline void smol_wait_vmcnt (void) {}
static inline void smol_wave_sync (void) { __syncthreads (); }
static inline uint32_t smol_wave_index (void) { return 0; }       /* emulated blocks are one wave */

/* v_readlane: every lane of the block calls this with the same idx */
extern uint32_t emu_lane_xchg [64];
static inline uint32_t smol_readlane (uint32_t v, uint32_t idx)
{
    emu_lane_xchg [threadIdx.x] = v;
    __syncthreads ();
    uint32_t r = emu_lane_xchg [idx];
    __syncthreads ();
    return r;
}

static inline uint32_t smol_shfl_xor (uint32_t v, uint32_t mask)
{
    emu_lane_xchg [threadIdx.x] = v;
    __syncthreads ();
    uint32_t r = emu_lane_xchg [threadIdx.x ^ mask];
    __syncthreads ();
    return r;
}

static inline uint32_t wave_sum (uint32_t v)
{
    for (uint32_t m = 32; m >= 1; m >>= 1)
        v += smol_shfl_xor (v, m);
    return v;
}

static inline uint32_t smol_shfl_idx (uint32_t v, uint32_t src)
{
    emu_lane_xchg [threadIdx.x] = v;
    __syncthreads ();
    uint32_t r = emu_lane_xchg [src & 63u];
    __syncthreads ();
    return r;
}

template <int P> static inline void smol_setprio (void) {}
