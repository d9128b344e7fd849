// hspan_ab.hip -- A/B of k_hspan code variants on BASELINE config 4 in ONE process (interleaved rounds,
// cdna_hip_programming.md rule 24), with the product's own headers: the JobDev is filled by the same
// host code the library uses. Build (from the repo root):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ismolscale_amd/csrc tools/microbench/hspan_ab.hip smolscale_amd/csrc/smol_plan.c -x hip -o tests/exp/hspan_ab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include "smol_kernels.hip.h"
#include "smol_kernel_tile.hip.h"
#include "smol_kernel_hspan.hip.h"
using namespace smolhip;
using namespace smolhip::dispatch;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e)); return 1; } } while (0)

template <int VAR, int WPE>
__global__ void __launch_bounds__ (256, WPE) k_var (const JobDev J)
{
    hspan_dispatch_out<3, 2, VAR> (J, smol_dyn_lds);
}

struct Variant { const char *name; void (*launch) (const JobDev &, dim3, uint32_t); std::vector<float> us; };
template <int VAR, int WPE> static void launch_v (const JobDev &J, dim3 grid, uint32_t lds)
{
    hipLaunchKernelGGL ((k_var<VAR, WPE>), grid, dim3 (256), lds, 0, J);
}

int main (int argc, char **argv)
{
    const uint32_t w_in = 65535, h = 4096, w_out = 8192;
    SmolHipJob job;
    if (smol_hip_plan_job (8, w_in, h, 0, w_out, h, 0, &job)) return 1;
    uint8_t *s, *d;
    const size_t in_b = (size_t) w_in * 3 * h + 64, out_b = (size_t) w_out * 4 * h;
    CK (hipMalloc (&s, in_b)); CK (hipMalloc (&d, out_b)); CK (hipMemset (s, 0x5a, in_b));
    hipEvent_t e0, e1; CK (hipEventCreate (&e0)); CK (hipEventCreate (&e1));
    std::vector<Variant> vs = {
        { "base (prefetch, lazy frames, barriers) wpe4", launch_v<0, 4>, {} },
        { "no prefetch                            wpe4", launch_v<1, 4>, {} },
        { "eager frames                           wpe4", launch_v<2, 4>, {} },
        { "no sched barrier                       wpe4", launch_v<4, 4>, {} },
        { "eager + no barrier                     wpe4", launch_v<6, 4>, {} },
        { "eager + no barrier + no prefetch       wpe4", launch_v<7, 4>, {} },
        { "base                                   wpe3", launch_v<0, 3>, {} },
        { "ABLATION no unpack/taps                wpe4", launch_v<8, 4>, {} },
        { "ABLATION no unpack/taps, no prefetch   wpe4", launch_v<9, 4>, {} },
        { "ABLATION no unpack/taps                wpe5", launch_v<8, 5>, {} },
    };
    for (uint32_t band : { 1u, 2u, 4u, 8u, 64u })
    {
        JobDev J;
        fill_jobdev (J, job, s, w_in * 3, d, w_out * 4, 0, h);
        if (classify (J) != KC_HSPAN) { printf ("not an hspan job\n"); return 1; }
        J.tile_rows = (uint16_t) band;
        const dim3 grid (hspan_blocks (J));
        const uint32_t lds = hspan_lds_bytes (J);
        for (auto &v : vs) v.us.clear ();
        for (int round = 0; round < 5; round++)
            for (auto &v : vs)
            {
                v.launch (J, grid, lds);
                (void) hipEventRecord (e0, 0);
                for (int i = 0; i < 10; i++) v.launch (J, grid, lds);
                (void) hipEventRecord (e1, 0); (void) hipEventSynchronize (e1);
                float ms; (void) hipEventElapsedTime (&ms, e0, e1);
                v.us.push_back (ms * 100.0f);
            }
        CK (hipGetLastError ());
        for (auto &v : vs)
        {
            std::sort (v.us.begin (), v.us.end ());
            printf ("band %2u  %-48s min %7.1f  median %7.1f us  (%.1f %% of 8 TB/s at median)\n", band, v.name, v.us.front (), v.us [v.us.size () / 2],
                    939511808.0 / v.us [v.us.size () / 2] / 1e3 / 80.0);
        }
        fflush (stdout);
    }
    return 0;
}
