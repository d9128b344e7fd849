/* host_path.c -- the drop-in API from HOST memory, in C, without Python in the way: T threads, each making a context
 * per frame (smol_scale_new + smol_scale_batch + smol_scale_destroy) on its own malloc'ed (or page-locked) frame.
 *   host_path [threads=1] [frames per thread=20] [pinned=0] [w_in h_in w_out h_out]
 * Build: gcc -O2 -pthread -I include tools/microbench/host_path.c -o tools/microbench/host_path -Lsmolscale_amd -lsmolscale_amd -Wl,-rpath,$PWD/smolscale_amd
 * With SMOL_HIP_TRACE=1 the library says where every strip's time went. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "smolscale.h"
#include "smol_hip.h"

static uint32_t W_IN = 4096, H_IN = 4096, W_OUT = 1024, H_OUT = 1024;
static int frames = 20, pinned = 0;

static pthread_barrier_t bar;
static double t_start;

static double now (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

static void *
worker (void *arg)
{
    (void) arg;
    const size_t n_in = (size_t) W_IN * H_IN * 4, n_out = (size_t) W_OUT * H_OUT * 4;
    void *src, *dst;
    if (pinned)
    {
        if (smol_hip_host_alloc (n_in, &src) || smol_hip_host_alloc (n_out, &dst)) abort ();
    }
    else
    {
        src = malloc (n_in);
        dst = malloc (n_out);
    }
    for (size_t i = 0; i < n_in; i += 4)
        *(uint32_t *) ((char *) src + i) = (uint32_t) (i * 2654435761u) & 0x7f7f7f7fu;
    memset (dst, 0, n_out);
    /* buffers exist and are touched: the clock starts when every thread is here */
    if (pthread_barrier_wait (&bar) == PTHREAD_BARRIER_SERIAL_THREAD)
        t_start = now ();
    pthread_barrier_wait (&bar);
    for (int f = 0; f < frames; f++)
    {
        SmolScaleCtx *c = smol_scale_new (src, SMOL_PIXEL_RGBA8_PREMULTIPLIED, W_IN, H_IN, W_IN * 4,
                                          dst, SMOL_PIXEL_RGBA8_PREMULTIPLIED, W_OUT, H_OUT, W_OUT * 4, 0);
        smol_scale_batch (c, 0, H_OUT);
        smol_scale_destroy (c);
    }
    return NULL;
}

int
main (int argc, char **argv)
{
    int threads = argc > 1 ? atoi (argv [1]) : 1;
    if (argc > 2) frames = atoi (argv [2]);
    if (argc > 3) pinned = atoi (argv [3]);
    if (argc > 7) { W_IN = atoi (argv [4]); H_IN = atoi (argv [5]); W_OUT = atoi (argv [6]); H_OUT = atoi (argv [7]); }
    pthread_t th [64];
    int keep = frames;
    pthread_barrier_init (&bar, NULL, 1);
    frames = 2; worker (NULL); frames = keep;            /* warm-up: lanes, rings */
    pthread_barrier_destroy (&bar);
    pthread_barrier_init (&bar, NULL, (unsigned) threads);
    for (int i = 0; i < threads; i++) pthread_create (&th [i], NULL, worker, NULL);
    for (int i = 0; i < threads; i++) pthread_join (th [i], NULL);
    double dt = now () - t_start;
    const double per = dt / ((double) threads * frames);
    printf ("{\"threads\": %d, \"pinned\": %d, \"ms_per_frame\": %.3f, \"Mpix_out_per_s\": %.1f, \"GBps\": %.1f}\n", threads, pinned,
            per * 1e3, W_OUT * (double) H_OUT / per / 1e6, ((double) W_IN * H_IN * 4 + (double) W_OUT * H_OUT * 4) / per / 1e9);
    return 0;
}
