// Experiment harness (not product code): variants of the 4:1 x 4:1 streaming kernel,
// interleaved rounds in one process (cdna_hip_programming.md 5.4 rule 24).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b) { return (a & b) + (((a ^ b) >> 1) & 0x7f7f7f7fu); }
__device__ __forceinline__ uint32_t red4(u32x4 v) { return avg4(avg4(v.x, v.y), avg4(v.z, v.w)); }

constexpr int W_IN = 4096, H_IN = 4096, W_OUT = 1024, H_OUT = 1024, NF = 8;

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(uint32_t* p, uint32_t v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <int POL> __device__ __forceinline__ void stp(uint32_t* p, uint32_t v)
{
    if (POL == 1) __builtin_nontemporal_store(v, p);
    else if (POL == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (POL == 3) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else *p = v;
}
// load-policy variants: the four loads as one asm statement (own waitcnt), stores write-through
#define LD4(POLSTR) asm volatile ( \
    "global_load_dwordx4 %0, %4, off " POLSTR "\n\tglobal_load_dwordx4 %1, %5, off " POLSTR "\n\t" \
    "global_load_dwordx4 %2, %6, off " POLSTR "\n\tglobal_load_dwordx4 %3, %7, off " POLSTR "\n\ts_waitcnt vmcnt(0)" \
    : "=&v"(raw[0]), "=&v"(raw[1]), "=&v"(raw[2]), "=&v"(raw[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory")
template <int LP>
__global__ void __launch_bounds__(1024) k_wrowl(const uint8_t* src, uint8_t* dst)
{
    const uint32_t tiles_x = W_OUT / 64;
    const uint32_t f = blockIdx.y;
    const uint32_t row = (blockIdx.x / tiles_x) * 16 + threadIdx.y;
    const uint32_t tc = (blockIdx.x % tiles_x) * 64 + threadIdx.x;
    const uint8_t* s = src + (size_t)f * W_IN * H_IN * 4 + (size_t)row * 4 * (W_IN * 4) + (size_t)tc * 16;
    const u32x4 *p0 = (const u32x4*)s, *p1 = (const u32x4*)(s + W_IN * 4), *p2 = (const u32x4*)(s + 2 * W_IN * 4), *p3 = (const u32x4*)(s + 3 * W_IN * 4);
    u32x4 raw[4];
    if (LP == 0) LD4("nt");
    else if (LP == 1) LD4("");
    else if (LP == 2) LD4("sc1");
    else if (LP == 3) LD4("sc0 sc1");
    else if (LP == 4) LD4("nt sc1");
    else if (LP == 5) LD4("nt sc0");
    else LD4("nt sc0 sc1");
    uint32_t* d = (uint32_t*)(dst + (size_t)f * W_OUT * H_OUT * 4 + (size_t)row * (W_OUT * 4)) + tc;
    stp<2>(d, avg4(avg4(red4(raw[0]), red4(raw[1])), avg4(red4(raw[2]), red4(raw[3]))));
}
template <int RB, int POL>
__global__ void __launch_bounds__(64 * RB) k_wrowp(const uint8_t* src, uint8_t* dst)
{
    const uint32_t tiles_x = W_OUT / 64;
    const uint32_t f = blockIdx.y;
    const uint32_t row = (blockIdx.x / tiles_x) * RB + threadIdx.y;
    const uint32_t tc = (blockIdx.x % tiles_x) * 64 + threadIdx.x;
    const uint8_t* s = src + (size_t)f * W_IN * H_IN * 4 + (size_t)row * 4 * (W_IN * 4) + (size_t)tc * 16;
    u32x4 raw[4];
#pragma unroll
    for (int r = 0; r < 4; r++) raw[r] = ld<true>((const u32x4*)(s + (size_t)r * (W_IN * 4)));
    uint32_t* d = (uint32_t*)(dst + (size_t)f * W_OUT * H_OUT * 4 + (size_t)row * (W_OUT * 4)) + tc;
    stp<POL>(d, avg4(avg4(red4(raw[0]), red4(raw[1])), avg4(red4(raw[2]), red4(raw[3]))));
}

// V0: thread = 1 output px, RPT output rows per thread; block BLOCK threads along x
template <int RPT, bool NTL, bool NTS, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_rows(const uint8_t* src, uint8_t* dst)
{
    const uint32_t tiles_x = W_OUT / BLOCK;
    const uint32_t f = blockIdx.y;
    const uint32_t row0 = (blockIdx.x / tiles_x) * RPT;
    const uint32_t tc = (blockIdx.x % tiles_x) * BLOCK + threadIdx.x;
    const uint8_t* s = src + (size_t)f * W_IN * H_IN * 4 + (size_t)row0 * 4 * (W_IN * 4) + (size_t)tc * 16;
    u32x4 raw[RPT * 4];
#pragma unroll
    for (int r = 0; r < RPT * 4; r++) raw[r] = ld<NTL>((const u32x4*)(s + (size_t)r * (W_IN * 4)));
    uint32_t* d = (uint32_t*)(dst + (size_t)f * W_OUT * H_OUT * 4 + (size_t)row0 * (W_OUT * 4)) + tc;
#pragma unroll
    for (int o = 0; o < RPT; o++) {
        uint32_t h0 = red4(raw[o*4+0]), h1 = red4(raw[o*4+1]), h2 = red4(raw[o*4+2]), h3 = red4(raw[o*4+3]);
        st<NTS>(d + (size_t)o * W_OUT, avg4(avg4(h0, h1), avg4(h2, h3)));
    }
}

// V3: block = 64 columns x RB output rows (one wave per output row)
template <int RB, bool NTS>
__global__ void __launch_bounds__(64 * RB) k_wrow(const uint8_t* src, uint8_t* dst)
{
    const uint32_t tiles_x = W_OUT / 64;
    const uint32_t f = blockIdx.y;
    const uint32_t row = (blockIdx.x / tiles_x) * RB + threadIdx.y;
    const uint32_t tc = (blockIdx.x % tiles_x) * 64 + threadIdx.x;
    const uint8_t* s = src + (size_t)f * W_IN * H_IN * 4 + (size_t)row * 4 * (W_IN * 4) + (size_t)tc * 16;
    u32x4 raw[4];
#pragma unroll
    for (int r = 0; r < 4; r++) raw[r] = ld<true>((const u32x4*)(s + (size_t)r * (W_IN * 4)));
    uint32_t* d = (uint32_t*)(dst + (size_t)f * W_OUT * H_OUT * 4 + (size_t)row * (W_OUT * 4)) + tc;
    st<NTS>(d, avg4(avg4(red4(raw[0]), red4(raw[1])), avg4(red4(raw[2]), red4(raw[3]))));
}

// V1: persistent grid-stride over (frame,row,tile) work items
template <bool NTL, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_persist(const uint8_t* src, uint8_t* dst, uint32_t n_items)
{
    const uint32_t tiles_x = W_OUT / BLOCK;
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const uint32_t f = it / (tiles_x * H_OUT);
        const uint32_t rem = it % (tiles_x * H_OUT);
        const uint32_t row = rem / tiles_x;
        const uint32_t tc = (rem % tiles_x) * BLOCK + threadIdx.x;
        const uint8_t* s = src + (size_t)f * W_IN * H_IN * 4 + (size_t)row * 4 * (W_IN * 4) + (size_t)tc * 16;
        u32x4 raw[4];
#pragma unroll
        for (int r = 0; r < 4; r++) raw[r] = ld<NTL>((const u32x4*)(s + (size_t)r * (W_IN * 4)));
        uint32_t h0 = red4(raw[0]), h1 = red4(raw[1]), h2 = red4(raw[2]), h3 = red4(raw[3]);
        ((uint32_t*)(dst + (size_t)f * W_OUT * H_OUT * 4 + (size_t)row * (W_OUT * 4)))[tc] = avg4(avg4(h0, h1), avg4(h2, h3));
    }
}

// V2: thread handles 2 adjacent output px (2 x 16B loads per row, lane stride 32B), 8-byte store
template <bool NTL>
__global__ void __launch_bounds__(256) k_two(const uint8_t* src, uint8_t* dst)
{
    const uint32_t tiles_x = (W_OUT / 2) / 256;
    const uint32_t f = blockIdx.y;
    const uint32_t row = blockIdx.x / tiles_x;
    const uint32_t tc = (blockIdx.x % tiles_x) * 256 + threadIdx.x;
    const uint8_t* s = src + (size_t)f * W_IN * H_IN * 4 + (size_t)row * 4 * (W_IN * 4) + (size_t)tc * 32;
    u32x4 raw[8];
#pragma unroll
    for (int r = 0; r < 4; r++) { raw[2*r] = ld<NTL>((const u32x4*)(s + (size_t)r * (W_IN * 4))); raw[2*r+1] = ld<NTL>((const u32x4*)(s + (size_t)r * (W_IN * 4)) + 1); }
    uint32_t a = avg4(avg4(red4(raw[0]), red4(raw[2])), avg4(red4(raw[4]), red4(raw[6])));
    uint32_t b = avg4(avg4(red4(raw[1]), red4(raw[3])), avg4(red4(raw[5]), red4(raw[7])));
    ((uint2*)(dst + (size_t)f * W_OUT * H_OUT * 4 + (size_t)row * (W_OUT * 4)))[tc] = make_uint2(a, b);
}

// pure copy reference: read 16B/lane, write 16B/lane (same bytes read as the scaler)
__global__ void __launch_bounds__(256) k_copy(const u32x4* src, u32x4* dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
// read-only reference: sum and conditionally store
__global__ void __launch_bounds__(256) k_read(const u32x4* src, uint32_t* dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256), c = __builtin_nontemporal_load(src + i + 512), d = __builtin_nontemporal_load(src + i + 768);
    uint32_t s = red4(a) ^ red4(b) ^ red4(c) ^ red4(d);
    if (s == 0x12345679u) dst[i] = s;
}

struct Var { const char* name; void (*launch)(const uint8_t*, uint8_t*, hipStream_t); };

int main()
{
    const size_t in_bytes = (size_t)NF * W_IN * H_IN * 4, out_bytes = (size_t)NF * W_OUT * H_OUT * 4;
    uint8_t *src, *dst, *dst_ref;
    CK(hipMalloc(&src, in_bytes)); CK(hipMalloc(&dst, out_bytes > in_bytes ? out_bytes : in_bytes)); CK(hipMalloc(&dst_ref, out_bytes));
    std::vector<uint32_t> h(in_bytes / 4);
    uint64_t x = 88172645463325252ull;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)x; }
    CK(hipMemcpy(src, h.data(), in_bytes, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<Var> vars = {
        {"rows RPT1 ntL  B256 (shipped)", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_rows<1,true,false,256>), dim3(W_OUT/256*H_OUT, NF), dim3(256), 0, s, a, b); }},
        {"rows RPT1 plainL B256", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_rows<1,false,false,256>), dim3(W_OUT/256*H_OUT, NF), dim3(256), 0, s, a, b); }},
        {"rows RPT1 ntL ntS B256", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_rows<1,true,true,256>), dim3(W_OUT/256*H_OUT, NF), dim3(256), 0, s, a, b); }},
        {"rows RPT2 ntL  B256", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_rows<2,true,false,256>), dim3(W_OUT/256*H_OUT/2, NF), dim3(256), 0, s, a, b); }},
        {"rows RPT4 ntL  B256", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_rows<4,true,false,256>), dim3(W_OUT/256*H_OUT/4, NF), dim3(256), 0, s, a, b); }},
        {"rows RPT1 ntL  B1024", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_rows<1,true,false,1024>), dim3(H_OUT, NF), dim3(1024), 0, s, a, b); }},
        {"rows RPT1 ntL  B64", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_rows<1,true,false,64>), dim3(W_OUT/64*H_OUT, NF), dim3(64), 0, s, a, b); }},
        {"rows RPT2 ntL  B1024", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_rows<2,true,false,1024>), dim3(H_OUT/2, NF), dim3(1024), 0, s, a, b); }},
        {"persist 2048 blocks ntL", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_persist<true,256>), dim3(2048), dim3(256), 0, s, a, b, (uint32_t)(NF*H_OUT*W_OUT/256)); }},
        {"persist 4096 blocks ntL", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_persist<true,256>), dim3(4096), dim3(256), 0, s, a, b, (uint32_t)(NF*H_OUT*W_OUT/256)); }},
        {"two px/thread ntL", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_two<true>), dim3(W_OUT/2/256*H_OUT, NF), dim3(256), 0, s, a, b); }},
        {"rows RPT1 ntL ntS B64", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_rows<1,true,true,64>), dim3(W_OUT/64*H_OUT, NF), dim3(64), 0, s, a, b); }},
        {"rows RPT1 ntL ntS B128", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_rows<1,true,true,128>), dim3(W_OUT/128*H_OUT, NF), dim3(128), 0, s, a, b); }},
        {"rows RPT1 ntL ntS B512", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_rows<1,true,true,512>), dim3(W_OUT/512*H_OUT, NF), dim3(512), 0, s, a, b); }},
        {"rows RPT2 ntL ntS B64", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_rows<2,true,true,64>), dim3(W_OUT/64*H_OUT/2, NF), dim3(64), 0, s, a, b); }},
        {"wrow 64x4 ntS", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrow<4,true>), dim3(W_OUT/64*H_OUT/4, NF), dim3(64,4), 0, s, a, b); }},
        {"wrow 64x8 ntS", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrow<8,true>), dim3(W_OUT/64*H_OUT/8, NF), dim3(64,8), 0, s, a, b); }},
        {"wrow 64x16 ntS", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrow<16,true>), dim3(W_OUT/64*H_OUT/16, NF), dim3(64,16), 0, s, a, b); }},
        {"wrowp 64x16 sc1 (agent) S", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrowp<16,2>), dim3(W_OUT/64*H_OUT/16, NF), dim3(64,16), 0, s, a, b); }},
        {"wrowp 64x16 sc0sc1 (system) S", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrowp<16,3>), dim3(W_OUT/64*H_OUT/16, NF), dim3(64,16), 0, s, a, b); }},
        {"wrowl loads nt       + sc1 stores", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrowl<0>), dim3(W_OUT/64*H_OUT/16, NF), dim3(64,16), 0, s, a, b); }},
        {"wrowl loads plain    + sc1 stores", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrowl<1>), dim3(W_OUT/64*H_OUT/16, NF), dim3(64,16), 0, s, a, b); }},
        {"wrowl loads sc1      + sc1 stores", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrowl<2>), dim3(W_OUT/64*H_OUT/16, NF), dim3(64,16), 0, s, a, b); }},
        {"wrowl loads sc0 sc1  + sc1 stores", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrowl<3>), dim3(W_OUT/64*H_OUT/16, NF), dim3(64,16), 0, s, a, b); }},
        {"wrowl loads nt sc1   + sc1 stores", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrowl<4>), dim3(W_OUT/64*H_OUT/16, NF), dim3(64,16), 0, s, a, b); }},
        {"wrowl loads nt sc0   + sc1 stores", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrowl<5>), dim3(W_OUT/64*H_OUT/16, NF), dim3(64,16), 0, s, a, b); }},
        {"wrowl loads nt sc0 sc1+sc1 stores", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrowl<6>), dim3(W_OUT/64*H_OUT/16, NF), dim3(64,16), 0, s, a, b); }},
        {"wrow 64x16 plainS", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrow<16,false>), dim3(W_OUT/64*H_OUT/16, NF), dim3(64,16), 0, s, a, b); }},
        {"wrow 64x8 plainS", [](const uint8_t* a, uint8_t* b, hipStream_t s){ hipLaunchKernelGGL((k_wrow<8,false>), dim3(W_OUT/64*H_OUT/8, NF), dim3(64,8), 0, s, a, b); }},
        {"REF read-only 16B x4 nt", [](const uint8_t* a, uint8_t* b, hipStream_t s){ size_t n = (size_t)NF*W_IN*H_IN*4/16; hipLaunchKernelGGL(k_read, dim3((unsigned)(n/4/256)), dim3(256), 0, s, (const u32x4*)a, (uint32_t*)b, n); }},
    };
    // reference output
    vars[0].launch(src, dst_ref, s); CK(hipStreamSynchronize(s));
    std::vector<uint32_t> ref(out_bytes / 4), got(out_bytes / 4);
    CK(hipMemcpy(ref.data(), dst_ref, out_bytes, hipMemcpyDeviceToHost));
    const int ROUNDS = 5, ITERS = 20;
    std::vector<std::vector<float>> t(vars.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < ROUNDS; r++)
        for (size_t v = 0; v < vars.size(); v++) {
            vars[v].launch(src, dst, s); vars[v].launch(src, dst, s);
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < ITERS; i++) vars[v].launch(src, dst, s);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t[v].push_back(ms / ITERS * 1000.f);
            if (r == 0 && v + 1 < vars.size()) {
                CK(hipMemcpy(got.data(), dst, out_bytes, hipMemcpyDeviceToHost));
                if (got != ref) printf("!! variant %s output differs\n", vars[v].name);
            }
        }
    const double bytes = (double)in_bytes + out_bytes;
    for (size_t v = 0; v < vars.size(); v++) {
        std::sort(t[v].begin(), t[v].end());
        double b = (v + 1 == vars.size()) ? (double)in_bytes : bytes;
        printf("%-34s min %7.2f us  med %7.2f us   %7.1f GB/s (med)  %5.1f%% of 8TB/s\n", vars[v].name, t[v][0], t[v][ROUNDS/2], b / t[v][ROUNDS/2] / 1e3, b / t[v][ROUNDS/2] / 1e3 / 80.0);
    }
    return 0;
}
