/* cpu_backend.c -- the library's CPU backend (smol_cpu.c) by job class, ms per frame and source Mpix/s; SMOL_CPU_THREADS sets the
 * strips' threads. No arguments: the table of classes below; or: t_in w_in h_in t_out w_out h_out [flags].
 * Build: gcc -O2 -I include tools/microbench/cpu_backend.c -o tools/microbench/cpu_backend -Lsmolscale_amd -lsmolscale_amd -Wl,-rpath,$PWD/smolscale_amd */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <time.h>
#include "smol_hip.h"
#include "smol_cpu.h"
static double now (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static int bpp (int t) { return t >= 8 ? 3 : 4; }
static int
one (const char *name, int ti, int wi, int hi, int to, int wo, int ho, int flags)
{
    SmolHipJob job;
    const size_t sp = (size_t) wi * bpp (ti), dp = (size_t) wo * bpp (to);
    if (smol_hip_plan_job (ti, wi, hi, to, wo, ho, flags, &job)) return 1;
    uint8_t *src = malloc (sp * hi), *dst = malloc (dp * ho);
    for (size_t i = 0; i < sp * hi; i++) src [i] = (uint8_t) (i * 7);
    smol_cpu_scale_rows (&job, src, sp, dst, dp, 0, ho);
    int reps = 0;
    double t = now (), e;
    do { smol_cpu_scale_rows (&job, src, sp, dst, dp, 0, ho); reps++; } while ((e = now () - t) < 0.5);
    printf ("%-44s %8.2f ms  %8.0f Mpix/s\n", name, e / reps * 1e3, (double) wi * hi * reps / e * 1e-6);
    free (src); free (dst);
    return 0;
}
int main (int argc, char **argv)
{
    if (argc >= 7)
        return one ("job", atoi (argv [1]), atoi (argv [2]), atoi (argv [3]), atoi (argv [4]), atoi (argv [5]), atoi (argv [6]),
                    argc > 7 ? atoi (argv [7]) : 0);
    /* pixel types: 0-3 premultiplied 8-bit, 4-7 unassociated, 8-9 24-bit (include/smolscale.h) */
    one ("RGBA8 premul 4096^2 -> 1024^2 (C2)", 0, 4096, 4096, 0, 1024, 1024, 0);
    one ("RGBA8 premul 2048^2 -> 1365^2", 0, 2048, 2048, 0, 1365, 1365, 0);
    one ("RGBA8 unassoc -> premul 2048^2 -> 1365^2", 4, 2048, 2048, 0, 1365, 1365, 0);
    one ("RGBA8 premul -> unassoc 2048^2 -> 1365^2", 0, 2048, 2048, 4, 1365, 1365, 0);
    one ("RGBA8 unassoc -> BGRA8 unassoc 2048^2 -> 1365^2", 4, 2048, 2048, 5, 1365, 1365, 0);
    one ("RGB8 -> RGB8 2000x1500 -> 800x600", 8, 2000, 1500, 8, 800, 600, 0);
    one ("RGB8 -> RGBA8 premul 2000x1500 -> 800x600", 8, 2000, 1500, 0, 800, 600, 0);
    one ("RGBA8 unassoc -> RGB8 2000x1500 -> 800x600", 4, 2000, 1500, 8, 800, 600, 0);
    one ("RGBA8 premul 2048^2 -> 224^2 (box)", 0, 2048, 2048, 0, 224, 224, 0);
    one ("RGB8 4000x3000 -> 400x300 (box)", 8, 4000, 3000, 8, 400, 300, 0);
    one ("RGBA8 premul 1024^2 -> 2048^2 (up)", 0, 1024, 1024, 0, 2048, 2048, 0);
    one ("RGBA8 unassoc sRGB 2048^2 -> 1365^2", 4, 2048, 2048, 4, 1365, 1365, 1);
    one ("RGB8 sRGB 2000x1500 -> 800x600", 8, 2000, 1500, 8, 800, 600, 1);
    one ("RGBA8 premul sRGB 4096^2 -> 448^2 (box)", 0, 4096, 4096, 0, 448, 448, 1);
    one ("RGB8 65535x256 -> RGBA8 8192x256 (C4 rows)", 8, 65535, 256, 0, 8192, 256, 0);
    return 0;
}
