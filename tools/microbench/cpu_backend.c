/* cpu_backend.c -- the library's CPU backend (smol_cpu.c) on a 4096^2 -> 1024^2 RGBA frame, ms per frame; SMOL_CPU_THREADS sets the strips' threads.
 * Build: gcc -O2 -I include tools/microbench/cpu_backend.c -o tools/microbench/cpu_backend -Lsmolscale_amd -lsmolscale_amd -Wl,-rpath,$PWD/smolscale_amd */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <time.h>
#include "smol_hip.h"
#include "smol_cpu.h"
static double now (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main (void)
{
    SmolHipJob job;
    if (smol_hip_plan_job (0, 4096, 4096, 0, 1024, 1024, 0, &job)) return 1;
    uint8_t *src = malloc (4096u * 4096u * 4u), *dst = malloc (1024u * 1024u * 4u);
    for (size_t i = 0; i < 4096u * 4096u * 4u; i++) src [i] = (uint8_t) (i * 7);
    smol_cpu_scale_rows (&job, src, 4096 * 4, dst, 1024 * 4, 0, 1024);
    double t = now ();
    for (int i = 0; i < 10; i++) smol_cpu_scale_rows (&job, src, 4096 * 4, dst, 1024 * 4, 0, 1024);
    printf ("%.2f ms per frame\n", (now () - t) * 100.0);
    return 0;
}
