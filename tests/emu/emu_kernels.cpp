/* emu_kernels.cpp -- TEST INFRASTRUCTURE: runs the HIP kernel bodies on the host,
 * block by block and thread by thread, through the same descriptor-filling and
 * kernel-class selection code the real runtime uses (smol_jobdev.h).
 * Built by tests/emu/Makefile with clang++ -DSMOL_KERNEL_EMU. */
#define SMOL_KERNEL_EMU 1
#include "smol_kernels_all.hip.h"

namespace smolhip {
/* k_lind under host emulation: one entry that switches at run time (the product has one entry point per flavour,
 * smol_k_updown.hip) */
__global__ void k_down_one (const JobDev J)
{
    const int in = J.p16 ? 2 : J.bpp_in == 3 ? 1 : 0;
    if (dispatch::lind_narrow (J))              /* tiles of at most 64 columns: the one-column flavour */
    {
        switch (J.halv_h * 3 + in)
        {
            case 3: lind_entry<1, false, false, true> (J, smol_dyn_lds); break;
            case 4: lind_entry<1, true, false, true> (J, smol_dyn_lds); break;
            case 5: lind_entry<1, false, true, true> (J, smol_dyn_lds); break;
            case 6: lind_entry<2, false, false, true> (J, smol_dyn_lds); break;
            case 7: lind_entry<2, true, false, true> (J, smol_dyn_lds); break;
            default: lind_entry<2, false, true, true> (J, smol_dyn_lds); break;
        }
        return;
    }
    switch (J.halv_h * 3 + in)
    {
        /* (no vertical halving: the flavour with the sample count compiled in, as the launcher picks it) */
        case 0: if (dispatch::lind_hv0 (J)) lind_entry<0, false, false, false, false, true> (J, smol_dyn_lds); else lind_entry<0, false, false> (J, smol_dyn_lds); break;
        case 1: if (dispatch::lind_hv0 (J)) lind_entry<0, true, false, false, false, true> (J, smol_dyn_lds); else lind_entry<0, true, false> (J, smol_dyn_lds); break;
        case 2: if (dispatch::lind_hv0 (J)) lind_entry<0, false, true, false, false, true> (J, smol_dyn_lds); else lind_entry<0, false, true> (J, smol_dyn_lds); break;
        case 3: lind_entry<1, false, false> (J, smol_dyn_lds); break;
        case 4: lind_entry<1, true, false> (J, smol_dyn_lds); break;
        case 5: lind_entry<1, false, true> (J, smol_dyn_lds); break;
        case 6: lind_entry<2, false, false> (J, smol_dyn_lds); break;
        case 7: lind_entry<2, true, false> (J, smol_dyn_lds); break;
        default: lind_entry<2, false, true> (J, smol_dyn_lds); break;
    }
}
} /* namespace smolhip */

#include <thread>
#include <vector>

#include "smol_rt_chunks.h"

pthread_barrier_t emu_block_barrier;
uint8_t *emu_dyn_lds_ptr;
uint32_t emu_lane_xchg [64];

/* cooperative kernels: the 64 lanes of a block run as real threads, blocks in turn */
template <typename F>
static void run_grid_coop (unsigned gx, unsigned bx, size_t lds_bytes, F body, unsigned gy = 1)
{
    std::vector<uint8_t> lds (lds_bytes + 64, 0xcd);
    emu_dyn_lds_ptr = (uint8_t *) (((uintptr_t) lds.data () + 15) & ~(uintptr_t) 15);
    pthread_barrier_init (&emu_block_barrier, nullptr, bx);
    for (unsigned g2 = 0; g2 < gy; g2++)
    for (unsigned b = 0; b < gx; b++)
    {
        std::vector<std::thread> lanes;
        for (unsigned t = 0; t < bx; t++)
            lanes.emplace_back ([=] {
                gridDim = { gx, gy, 1 };
                blockDim = { bx, 1, 1 };
                blockIdx = { b, g2, 0 };
                threadIdx = { t, 0, 0 };
                body ();
            });
        for (auto &th : lanes)
            th.join ();
    }
    pthread_barrier_destroy (&emu_block_barrier);
}

using namespace smolhip;
using namespace smolhip::dispatch;

template <typename F>
static void run_grid (unsigned gx, unsigned gy, unsigned bx, F body)
{
    gridDim = { gx, gy, 1 };
    blockDim = { bx, 1, 1 };
    for (unsigned by = 0; by < gy; by++)
        for (unsigned b = 0; b < gx; b++)
            for (unsigned t = 0; t < bx; t++)
            {
                blockIdx = { b, by, 0 };
                threadIdx = { t, 0, 0 };
                body ();
            }
}

template <typename F>
static void run_grid_2d (unsigned gx, unsigned bx, unsigned by, F body, unsigned gy = 1)
{
    gridDim = { gx, gy, 1 };
    blockDim = { bx, by, 1 };
    for (unsigned g2 = 0; g2 < gy; g2++)
        for (unsigned b = 0; b < gx; b++)
            for (unsigned ty = 0; ty < by; ty++)
                for (unsigned t = 0; t < bx; t++)
                {
                    blockIdx = { b, g2, 0 };
                    threadIdx = { t, ty, 0 };
                    body ();
                }
}

template <int HX>
static void run_pow2_hy (const JobDev &J, unsigned gx, unsigned gy)
{
    switch (J.halv_v)
    {
        case 0: run_grid_2d (gx, 64, 16, [&] { k_pow2_one<HX, 0> (J, smolhip::TileSwz { 0, 0, 0, 0, 0, 0, 0 }); }, gy); break;
        case 1: run_grid_2d (gx, 64, 16, [&] { k_pow2_one<HX, 1> (J, smolhip::TileSwz { 0, 0, 0, 0, 0, 0, 0 }); }, gy); break;
        default: run_grid_2d (gx, 64, 16, [&] { k_pow2_one<HX, 2> (J, smolhip::TileSwz { 0, 0, 0, 0, 0, 0, 0 }); }, gy); break;
    }
}

template <int H>
static void run_pow2x_h (const JobDev &J, unsigned gx, unsigned gy)
{
    switch (J.px_kind)
    {
        case PX_N32: run_grid_2d (gx, 64, 16, [&] { k_pow2x_one<H, PX_N32> (J); }, gy); break;
        case PX_N24: run_grid_2d (gx, 64, 16, [&] { k_pow2x_one<H, PX_N24> (J); }, gy); break;
        case PX_W32: run_grid_2d (gx, 64, 16, [&] { k_pow2x_one<H, PX_W32> (J); }, gy); break;
        case PX_W16: run_grid_2d (gx, 64, 16, [&] { k_pow2x_one<H, PX_W16> (J); }, gy); break;
        default:     run_grid_2d (gx, 64, 16, [&] { k_pow2x_one<H, PX_W24> (J); }, gy); break;
    }
}

extern "C" int
emu_scale_rows (const SmolHipJob *job, const void *src, uint32_t stride_in,
                void *dst, uint32_t stride_out, uint32_t first_row, uint32_t n_rows,
                int force_general, int *kclass_out)
{
    JobDev J;
    fill_jobdev (J, *job, src, stride_in, dst, stride_out, first_row, n_rows);
    /* classify() adapts the descriptor to the class it picks; forcing another class below
     * must start from the pristine descriptor again */
    const JobDev pristine = J;
    KClass k = classify (J);
    if (force_general && (k == KC_POW2 || k == KC_POW2X || k == KC_UP || k == KC_DOWN || k == KC_BOX2 || k == KC_HSPAN || k == KC_TILE || k == KC_WTILE || k == KC_WAVEPX || k == KC_WAVEPX_WIDE))
        k = J.wide ? KC_GENERAL_WIDE : KC_GENERAL;
    if (force_general == 2 && tile_ok (J))
    {
        k = KC_TILE;                      /* force the tiled kernel where the fast path would win */
        J = pristine;
    }
    else if (force_general && (k == KC_GENERAL || k == KC_GENERAL_WIDE))
        J = pristine;
    if (kclass_out)
        *kclass_out = (int) k;

    if (k == KC_POW2)
    {
        const uint32_t rx = 2u << J.halv_h;
        const uint32_t opt = rx >= 4 ? 1 : 4 / rx;
        const uint32_t tcols = J.w_out / opt;
        const unsigned gx = (tcols + 63) / 64, gy = (n_rows + 15) / 16;
        switch (J.halv_h)
        {
            case 0: run_pow2_hy<0> (J, gx, gy); break;
            case 1: run_pow2_hy<1> (J, gx, gy); break;
            default: run_pow2_hy<2> (J, gx, gy); break;
        }
        return 0;
    }

    if (k == KC_POW2X)
    {
        const unsigned gx = pow2x_tiles_x (J), gy = pow2x_tiles_y (J);
        switch (J.halv_h)
        {
            case 0: run_pow2x_h<0> (J, gx, gy); break;
            case 1: run_pow2x_h<1> (J, gx, gy); break;
            default: run_pow2x_h<2> (J, gx, gy); break;
        }
        return 0;
    }

    if (k == KC_BOX2)
    {
        std::vector<uint8_t> tmp (box2_tmp_bytes (J) + 64);
        J.tmp = tmp.data ();
        run_grid_coop (box2_h_tiles (J), 64, box2_h_lds_bytes (J), [&] { k_boxh_one (J); });
        run_grid_coop (box2_v_tiles (J), 64, 0, [&] { k_boxv_one (J); });
        return 0;
    }

    if (k == KC_HSPAN)
    {
        /* 4 waves per workgroup, each with its private LDS image */
        run_grid_coop (hspan_blocks (J), 64 * HSPAN_WAVES, hspan_lds_bytes (J), [&] { k_hspan_one (J); });
        return 0;
    }

    if (k == KC_DOWN)
    {
        run_grid_coop (down_blocks (J), 64, down_lds_bytes (J), [&] { k_down_one (J); });
        return 0;
    }

    if (k == KC_UP)
    {
        run_grid_coop (up_blocks (J), 64, up_lds_bytes (J), [&] { k_up_one (J); });
        return 0;
    }

    if (k == KC_WAVEPX || k == KC_WAVEPX_WIDE)
    {
        if (k == KC_WAVEPX_WIDE)
            run_grid_coop (J.w_out * n_rows, 64, 0, [&] { k_wavepx_one<true> (J); });
        else
            run_grid_coop (J.w_out * n_rows, 64, 0, [&] { k_wavepx_one<false> (J); });
        return 0;
    }

    if (k == KC_WTILE)
    {
        const uint32_t tiles_x = (J.w_out + TILE_COLS - 1) / TILE_COLS;
        const uint32_t tiles_y = (n_rows + J.tile_rows - 1) / J.tile_rows;
        run_grid_coop (tiles_x * tiles_y, 64, tile_lds_bytes (J), [&] { k_wtile_one (J); });
        return 0;
    }

    if (k == KC_TILE)
    {
        const uint32_t tw = TILE_COLS * J.tile_cpl / J.tile_lpc;
        const uint32_t tiles_x = (J.w_out + tw - 1) / tw;
        const uint32_t tiles_y = (n_rows + J.tile_rows - 1) / J.tile_rows;
        run_grid_coop (tiles_x, 64, tile_lds_bytes (J), [&] { k_tile_one (J); }, tiles_y);
        return 0;
    }

    unsigned gx = ((J.w_out + GENERAL_BLOCK - 1) / GENERAL_BLOCK) * n_rows;
    if (k == KC_GENERAL_WIDE)
        run_grid (gx, 1, GENERAL_BLOCK, [&] { k_general_one<true> (J); });
    else
        run_grid (gx, 1, GENERAL_BLOCK, [&] { k_general_one<false> (J); });
    return 0;
}

/* lerp11_pk (smol_kernel_tile.hip.h) against the reference's tap formula over EVERY (p, q, F) of the sRGB-linear
 * class: 11-bit p and q in the low lane (the high lane runs a permuted pair), F in 0..256. Returns the number of
 * mismatches. */
/* The host road's rule for strips cut into chunks (smol_rt_lanes.hip: strip_classify, then JobDev::box_force on every
 * chunk), replayed without a GPU: returns 0 when every chunk of `rows_per_chunk' rows takes the road the strip's probe
 * chose and fits the intermediate image the probe sized; 1 + the first offending chunk otherwise. `use_force' = 0 shows
 * what the chunks would decide on their own (the round-4 behaviour). *strip_class / *flips report what was seen. */
extern "C" int
emu_strip_chunks_agree (const SmolHipJob *job, uint32_t src_pitch, uint32_t dst_pitch, uint32_t first_row, uint32_t n_rows,
                        uint32_t rows_per_chunk, int use_force, int *strip_class, int *flips)
{
    const StripClass sc = strip_classify (*job, src_pitch, dst_pitch, first_row, n_rows);
    int bad = 0, n_flips = 0;

    if (strip_class)
        *strip_class = (int) sc.k;
    for (uint32_t done = 0, c = 0; done < n_rows; done += rows_per_chunk, c++)
    {
        const uint32_t rows = n_rows - done < rows_per_chunk ? n_rows - done : rows_per_chunk;
        JobDev jd;
        fill_jobdev (jd, *job, reinterpret_cast<const void *> (uintptr_t (16)), src_pitch,
                     reinterpret_cast<void *> (uintptr_t (16)), dst_pitch, first_row + done, rows);
        if (use_force)
            strip_force (jd, sc);
        const KClass k = classify (jd);
        const bool flip = (k == KC_BOX2) != (sc.k == KC_BOX2) || (k == KC_BOX2 && box2_tmp_bytes (jd) > sc.tmp_bytes);
        if (flip)
        {
            n_flips++;
            if (!bad)
                bad = 1 + (int) c;
        }
    }
    if (flips)
        *flips = n_flips;
    return bad;
}

/* The mapped road's chunk plan (smol_rt_chunks.h), replayed without a GPU: returns the output rows per chunk (0: no plan) and
 * checks what the road relies on -- every chunk's ring side(s) fit a slot, the chunks tile the row range, and every chunk's source
 * rows lie inside the strip's. < 0: a violated property. */
extern "C" int
emu_mapped_chunks (const SmolHipJob *job, uint32_t first_row, uint32_t n_rows, uint32_t src_pitch, uint32_t dst_pitch,
                   int ring_in, int ring_out, uint64_t want, uint64_t slot_max, uint64_t *in_max, uint64_t *out_max)
{
    const uint32_t bi = job->pixel_type_in >= 8 ? 3u : 4u, bo = job->pixel_type_out >= 8 ? 3u : 4u;
    const smolrt::MappedChunks mc = smolrt::mapped_chunks (*job, first_row, n_rows, (size_t) job->width_in * bi, (size_t) job->width_out * bo,
                                                           src_pitch, dst_pitch, ring_in != 0, ring_out != 0, want, slot_max);
    uint32_t r0 = 0, r1 = 0;

    if (in_max) *in_max = mc.in_max;
    if (out_max) *out_max = mc.out_max;
    if (!mc.rows)
        return 0;
    if (smol_hip_input_row_range (job, first_row, n_rows, &r0, &r1))
        return -1;
    if (!ring_in && !ring_out)
        return mc.rows == n_rows ? (int) mc.rows : -2;
    for (uint32_t y = 0; y < n_rows; y += mc.rows)
    {
        const uint32_t rows = n_rows - y < mc.rows ? n_rows - y : mc.rows;
        uint32_t a = 0, b = 0;
        if (smol_hip_input_row_range (job, first_row + y, rows, &a, &b))
            return -3;
        if (a < r0 || b > r1)
            return -4;
        if (ring_in && ((uint64_t) b - a + 1) * src_pitch > slot_max)
            return -5;
        if (ring_out && (uint64_t) rows * dst_pitch > slot_max)
            return -6;
    }
    return (int) mc.rows;
}

extern "C" long
emu_lerp11_exhaustive (void)
{
    using namespace smolhip;
    long bad = 0;

    for (uint32_t F = 0; F <= 256; F++)
    {
        const Lerp11W w = lerp11_weights (F);

        for (uint32_t p = 0; p < 2048; p++)
            for (uint32_t q = 0; q < 2048; q++)
            {
                const uint32_t p2 = 2047u - p, q2 = (q * 7u + 3u) & 2047u;
                const uint32_t r = lerp11_pk (p | (p2 << 16), q | (q2 << 16), w);
                const int64_t e0 = ((((int64_t) p - (int64_t) q) * (int64_t) F) >> 8) + (int64_t) q;
                const int64_t e1 = ((((int64_t) p2 - (int64_t) q2) * (int64_t) F) >> 8) + (int64_t) q2;

                if ((r & 0xffffu) != (uint32_t) e0 || (r >> 16) != (uint32_t) e1)
                    bad++;
            }
    }
    return bad;
}

/* ---- smol_hip_memory_zones()'s labelling rules (smol_zones_label.h) against a simulated device ------------------------------------
 * truth [w] = the kind of window w. Two streams of one kind take base x (0.985 .. 1.025), of two kinds base x (other_min .. other_min + 0.05), the halves
 * of one window base x (0.995 .. 1.005) -- those of window 0 x first_fast (the device: 0.975) --, and with probability p_disturb a
 * measurement is slowed by up to another 8 % (a disturbance only ever adds). */
#include "smol_zones_label.h"

extern "C" int
emu_zones_label (uint32_t n, const uint8_t *truth, uint32_t seed, float p_disturb, float first_fast, float other_min,
                 uint8_t *zones, uint32_t *kinds, float *contrast, uint32_t *n_measurements)
{
    uint64_t state = 0x9e3779b97f4a7c15ull ^ ((uint64_t) seed << 17);
    auto uni = [&] () -> float
    {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        return (float) ((state >> 40) & 0xffffff) / 16777216.0f;
    };
    uint32_t count = 0;
    auto pair = [&] (uint64_t a, uint64_t b, float *ms) -> int
    {
        const float base = 0.076f;
        float t;
        if (a == b)
            t = base * (a == 0 ? first_fast : 1.0f) * (0.995f + 0.01f * uni ());
        else if (truth [a] == truth [b])
            t = base * (0.985f + 0.04f * uni ());
        else
            t = base * (other_min + 0.05f * uni ());
        if (uni () < p_disturb)
            t *= 1.0f + 0.08f * uni ();
        *ms = t;
        count++;
        return 0;
    };
    const int rc = smolzones::label_windows (n, pair, zones, kinds, contrast, false);
    if (n_measurements)
        *n_measurements = count;
    return rc;
}
