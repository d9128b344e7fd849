/* smol_cpu.h -- C-ABI of the library's CPU backend: a from-scratch implementation of the same
 * resampling path, bit-exact with the reference's generic code (smolscale-generic.c:185-2318), kept as a FALLBACK:
 * AVX2 / AVX-512BW (chosen at run time) in every job class -- the 64bpp class in packed 16-bit lanes (the reference's
 * AVX2 speed per thread in the BASELINE class), the 128bpp class in 32-bit lanes --, plain 32-bit-lane code on CPUs
 * without AVX2 and under SMOL_CPU_SIMD=0; a call of enough rows is cut into strips of output rows over
 * the host cores (SMOL_CPU_THREADS, default the online cores up to 24; 1 = stay on the calling thread). It is what
 * smol_scale_*() runs on when the caller selects it with SMOL_BACKEND=cpu (or SMOL_BACKEND=auto on a machine without
 * a usable GPU); the default backend stays HIP, and a missing GPU then stays a loud failure. */

#ifndef SMOLSCALE_AMD_SMOL_CPU_H
#define SMOLSCALE_AMD_SMOL_CPU_H

#include <stdint.h>
#include "smol_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Rows [first_row, first_row + n_rows) of the job planned by smol_hip_plan_job(), host pointers,
 * row `first_row' written at dest + 0, rowstride_out apart (bytes between rows untouched).
 * Thread-safe. Returns 0 or SMOL_HIP_E_INVALID. */
int smol_cpu_scale_rows (const SmolHipJob *job,
                         const void *pixels_in, uint32_t rowstride_in,
                         void *dest, uint32_t rowstride_out,
                         uint32_t first_row, uint32_t n_rows);

/* The same with the per-column tap tables built once: a context that hands out its rows in many small
 * batches (the reference's row loop, smolscale.c:507-546) pays the O(width_out << halvings) set-up at
 * creation, not per call. The plan is immutable after creation: any number of threads may scale rows
 * through one plan at the same time. create returns SMOL_HIP_E_INVALID for a job that smol_hip_plan_job()
 * could not have produced (filter / halving / pixel type out of range, a bilinear axis of one pixel). */
typedef struct SmolCpuPlan SmolCpuPlan;
int smol_cpu_plan_create (const SmolHipJob *job, SmolCpuPlan **plan_out);
int smol_cpu_plan_scale_rows (const SmolCpuPlan *plan,
                              const void *pixels_in, uint32_t rowstride_in,
                              void *dest, uint32_t rowstride_out,
                              uint32_t first_row, uint32_t n_rows);
void smol_cpu_plan_destroy (SmolCpuPlan *plan);

#ifdef __cplusplus
}
#endif

#endif /* SMOLSCALE_AMD_SMOL_CPU_H */
