/* asan_driver.c -- TEST INFRASTRUCTURE. Sanitizer run (-fsanitize=address,undefined, CPU container only) over
 * the host-side product code: the planner (smol_plan.c), the CPU backend (smol_cpu.c), the public API
 * (smolscale_api.c, SMOL_BACKEND=cpu) and -- in the `emu' build -- the kernel bodies compiled for the host
 * (tests/emu). Every case is also compared byte for byte with the oracle (oracle/smol_oracle.c, built with
 * the same sanitizers), so this is a parity run and a memory-safety run in one.
 *
 *   asan_cpu  <cases> <seed>     planner + CPU backend + public API
 *   asan_emu  <cases> <seed>     emulated HIP kernel bodies (smaller images: lanes are threads)
 *
 * the reference defines TEST_DEBUG_CFLAGS (Makefile:28) and never uses it; this is the
 * sanitizer target it lacks. */

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smol_hip.h"
#include "smol_cpu.h"
#include "smolscale.h"
#include "smol_oracle.h"

#ifdef ASAN_EMU
int emu_scale_rows (const SmolHipJob *job, const void *src, uint32_t stride_in,
                    void *dst, uint32_t stride_out, uint32_t first_row, uint32_t n_rows,
                    int force_general, int *kclass_out);
#endif

static uint64_t rng_state;

static uint32_t
rnd (uint32_t n)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t) ((rng_state >> 11) % n);
}

static uint32_t bpp_of (uint32_t t) { return t >= 8 ? 3 : 4; }

/* output size for one axis: every filter class of the planner gets its share */
static void
pick_axis (uint32_t max_in, uint32_t *in, uint32_t *out)
{
    switch (rnd (8))
    {
        case 0: *in = 1; *out = 1 + rnd (40); break;                                    /* ONE       */
        case 1: *in = 1 + rnd (max_in); *out = *in; break;                              /* COPY      */
        case 2: *in = 2 + rnd (max_in / 4 + 1); *out = *in + 1 + rnd (3 * *in); break;  /* magnify   */
        case 3: *out = 1 + rnd (max_in / 2 + 1); *in = *out + 1 + rnd (*out); break;    /* 0H        */
        case 4: *out = 1 + rnd (max_in / 4 + 1); *in = 2 * *out + 1 + rnd (2 * *out); break;   /* 1H */
        case 5: *out = 1 + rnd (max_in / 8 + 1); *in = 4 * *out + 1 + rnd (4 * *out); break;   /* 2H */
        case 6: *out = 1 + rnd (max_in / 24 + 1); *in = 8 * *out + 1 + rnd (16 * *out); break; /* box */
        default: *out = 1 + rnd (3); *in = 256 * *out + rnd (64); break;                /* box > 255 */
    }
    if (*in > 65535) *in = 65535;
    if (*in < 1) *in = 1;
}

int
main (int argc, char **argv)
{
    const uint32_t n_cases = argc > 1 ? (uint32_t) atoi (argv [1]) : 100;
    uint32_t it, done = 0, bad = 0;

    rng_state = 0x9e3779b97f4a7c15ull ^ (argc > 2 ? (uint64_t) atoll (argv [2]) : 1);
#ifndef ASAN_EMU
    setenv ("SMOL_BACKEND", "cpu", 1);
#endif

    for (it = 0; done < n_cases && it < 100 * n_cases; it++)
    {
        uint32_t w_in, w_out, h_in, h_out;
#ifdef ASAN_EMU
        const uint32_t max_in = 700, max_px_in = 90000, max_px_out = 5000;
#else
        const uint32_t max_in = 3000, max_px_in = 700000, max_px_out = 300000;
#endif
        pick_axis (max_in, &w_in, &w_out);
        pick_axis (max_in, &h_in, &h_out);
        if ((uint64_t) w_in * h_in > max_px_in || (uint64_t) w_out * h_out > max_px_out)
            continue;
        const uint32_t t_in = rnd (10), t_out = rnd (10), srgb = rnd (4) == 0;
        const uint32_t s_in = w_in * bpp_of (t_in) + (rnd (2) ? 0 : rnd (9));
        const uint32_t s_out = w_out * bpp_of (t_out) + (rnd (2) ? 0 : rnd (9));
        const uint32_t mis = rnd (2) ? 0 : rnd (8);
        const uint32_t first = rnd (h_out), n = 1 + rnd (h_out - first);
        const size_t n_src = (size_t) s_in * h_in, n_dst = (size_t) s_out * n;
        /* exact-size heap blocks: a read or write one byte outside the image is a sanitizer report. The
         * kernels, by design, fetch whole 16-byte ALIGNED chunks that hold at least one byte of the image
         * (global_load_dwordx4 / LDS-DMA: such a chunk cannot cross a page, so it cannot fault on the device);
         * for them the block carries 16 bytes of margin on either side -- still a report for anything wider. */
#ifdef ASAN_EMU
        const size_t margin = 16;
#else
        const size_t margin = 0;
#endif
        uint8_t *src_block = malloc (n_src + mis + 2 * margin), *src = src_block + margin + mis;
        uint8_t *want = malloc (n_dst), *got = malloc (n_dst), *got2 = malloc (n_dst);
        SmolHipJob job;

        if (!src_block || !want || !got || !got2)
            return 2;
        smol_oracle_fill (src, n_src, it + 1);
        if (rnd (2))
            smol_oracle_make_valid_premul (src, t_in, w_in, h_in, s_in);
        memset (want, 0x5a, n_dst); memset (got, 0x5a, n_dst); memset (got2, 0x5a, n_dst);

        if (smol_hip_plan_job (t_in, w_in, h_in, t_out, w_out, h_out, srgb, &job))
        {
            fprintf (stderr, "planner rejected %u %ux%u -> %u %ux%u\n", t_in, w_in, h_in, t_out, w_out, h_out);
            return 3;
        }
        if (smol_oracle_scale_rows (src, t_in, w_in, h_in, s_in, want, t_out, w_out, h_out, s_out, srgb, first, n))
            return 4;
#ifdef ASAN_EMU
        {
            int kclass = -1;
            if (emu_scale_rows (&job, src, s_in, got, s_out, first, n, (int) rnd (6) == 0, &kclass))
                return 5;
            memcpy (got2, got, n_dst);
        }
#else
        {
            /* the C-ABI of the CPU backend, in two strips through one plan ... */
            SmolCpuPlan *plan;
            const uint32_t n_a = n / 2;
            uint32_t ra, rb;
            if (smol_cpu_plan_create (&job, &plan))
                return 5;
            if (smol_cpu_plan_scale_rows (plan, src, s_in, got, s_out, first, n_a)
                || smol_cpu_plan_scale_rows (plan, src, s_in, got + (size_t) n_a * s_out, s_out, first + n_a, n - n_a))
                return 6;
            smol_cpu_plan_destroy (plan);
            /* ... the row-range helper ... */
            if (smol_hip_input_row_range (&job, first, n, &ra, &rb) || ra > rb || rb >= h_in)
                return 7;
            /* ... and the public API on the same rows */
            SmolScaleCtx *ctx = smol_scale_new_full (src, (SmolPixelType) t_in, w_in, h_in, s_in,
                                                     NULL, (SmolPixelType) t_out, w_out, h_out, s_out,
                                                     srgb, NULL, NULL);
            if (!ctx)
                return 8;
            smol_scale_batch_full (ctx, got2, first, n);
            smol_scale_destroy (ctx);
        }
#endif
        /* bytes between rows must be untouched, bytes of the rows equal to the oracle's */
        if (memcmp (got, want, n_dst) || memcmp (got2, want, n_dst))
        {
            fprintf (stderr, "MISMATCH case %u: %u %ux%u (stride %u) -> %u %ux%u (stride %u) srgb %u rows %u+%u\n",
                     it, t_in, w_in, h_in, s_in, t_out, w_out, h_out, s_out, srgb, first, n);
            bad++;
        }
        free (src_block); free (want); free (got); free (got2);
        done++;
    }
    printf ("%s: %u cases, %u mismatches, no sanitizer report\n",
#ifdef ASAN_EMU
            "asan_emu",
#else
            "asan_cpu",
#endif
            done, bad);
    return bad ? 1 : 0;
}
