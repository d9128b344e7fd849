/* smol_jobdev.h -- the device job descriptor and the host-side logic that fills
 * it and picks a kernel class. Plain C++ (no HIP headers) so that the host
 * emulation harness in tests/emu can share it with the runtime (smol_rt_*.hip). */

#pragma once

#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#include "smol_hip.h"

/* the few helpers below that device code shares with the host */
#if defined (__HIPCC__) && !defined (SMOL_KERNEL_EMU)
#define SMOL_HD __host__ __device__
#else
#define SMOL_HD
#endif

namespace smolhip {

/* Tuning / ablation knobs (SMOL_TILE_ROWS, SMOL_NO_UP, SMOL_TILE_DBG ...) exist only in
 * builds made with -DSMOL_TUNING (make EXTRA=-DSMOL_TUNING, used by tools/
 * sweeps). The shipped library never looks at the environment on the dispatch path, so a
 * stray variable cannot change which kernel runs or what it computes. */
#ifdef SMOL_TUNING
inline const char *tune_env (const char *name) { return getenv (name); }
#else
inline const char *tune_env (const char *) { return nullptr; }
#endif

enum : uint8_t { F_COPY = 0, F_ONE = 1, F_BILINEAR = 2, F_BOX = 3 };

/* k_tile: vertical mode compiled into the entry point (JobDev::tile_vm), and the most vertical
 * samples a tile may hold (its schedule is one register: lane k = sample k, lane 63 a sentinel) */
enum : uint8_t { TILE_VM_ANY = 0, TILE_VM_COPY = 1, TILE_VM_PAIR = 2, TILE_VM_BOX = 3 };
constexpr uint32_t TILE_MAX_VSAMPLES = 63;

/* LDS copy of the parity tables: four inverse tables (4 x 1 KB), from_srgb (512 B), to_srgb (2 KB)
 * and the fused opaque-pixel linearisation (512 B); layout in smol_kernels.hip.h (LutLds) */
constexpr uint32_t SMOL_LUT_LDS_BYTES = 4 * 1024 + 512 + 2048 + 512;

/* k_pow2x input kinds: 32/24-bit input x 64bpp/128bpp-class math */
enum : int { PX_N32 = 0, PX_N24 = 1, PX_W32 = 2, PX_W24 = 3, PX_W16 = 4 /* 32-bit in, premul16 without sRGB */ };

/* One job as the kernels see it (device-resident for batches, by-value kernel
 * argument for single launches). Channel numbering: 0,1,2 = R,G,B, 3 = A. */
struct JobDev
{
    const uint8_t *src;            /* address of input row 0 (may lie before a row slab) */
    uint8_t *dst;                  /* address of output row `first_row'                  */
    uint8_t *tmp;                  /* k_boxh/k_boxv: intermediate image (rows tmp_row0..tmp_row1 x w_out x 8 bytes) */
    uint32_t src_stride, dst_stride;
    uint32_t w_in, h_in, w_out, h_out;
    uint32_t first_row, n_rows;
    uint32_t tmp_row0, tmp_row1;   /* k_boxh/k_boxv: first and last source row of the intermediate image */
    uint64_t frac0_x, step_x, frac0_y, step_y;
    uint32_t span_mul_x, span_mul_y;
    uint32_t out_perm;             /* v_perm selector, 32bpp -> 32bpp byte shuffle       */
    uint32_t up_perm;              /* k_up: out_perm composed with the gather of the four result bytes */
    uint8_t filter_h, filter_v, halv_h, halv_v;
    uint8_t bpp_in, bpp_out;
    uint8_t pos_in [4], pos_out [4];   /* byte index of R,G,B,A; 0xff = no alpha         */
    uint8_t unassoc_in, unassoc_out, wide, p16, linear, quirk24;
    uint8_t in_dword_ok, out_dword_ok; /* 4-byte pixels at 4-byte aligned addresses      */
    /* k_tile geometry (filled by tile_setup) */
    uint8_t tile_aligned4;         /* every staged pixel sits on a 4-byte LDS boundary   */
    uint8_t tile_variant;          /* tapsel << 4 | HM << 2 | OUT (smol_kernel_tile.hip.h) */
    uint8_t tile_cpl;              /* adjacent output columns per lane (1, 2 or 4)       */
    uint8_t tile_lpc;              /* lanes per output column (box: 1 or 8; k_boxh 128bpp class: also 16, 64) */
    uint8_t tile_band;             /* k_tile: deal tiles out in XCD bands (very wide images) */
    uint8_t tile_vm;               /* k_tile: vertical specialisation (TILE_VM_*): 0 decided at run time, 1 copy, 2 bilinear, 3 box */
    uint8_t out_vec16_ok;          /* 4 output pixels of a lane may go out as one 16-B store */
    uint8_t out3_ok;               /* 24-bit output whose full-width tiles leave as aligned dwords (any row alignment) */
    uint8_t px_kind;               /* k_pow2x: PX_* input kind                           */
    uint8_t px_lpr_shift;          /* k_pow2x: log2 (lanes of a wave per output row)     */
    uint8_t tile_single;           /* k_tile: one staging buffer, no prefetch of the next row group (more resident waves) */
    uint8_t box2_ygroup;           /* k_boxh: a workgroup's 4 waves take 4 row groups of one column tile (else 4 column tiles of one row group) */
    uint8_t px_mixed;              /* k_pow2x: member of a launch that mixes input kinds (batches only): the entry point switches per workgroup */
    uint8_t up_wide;               /* k_up: 0 64bpp class, 1 premul16 fast path, 2 table-driven 16-bit channels, 3 table-driven 32-bit */
    uint8_t box2;                  /* this job runs as k_boxh + k_boxv (two passes)     */
    uint8_t box2_vlpc;             /* k_boxv: lanes per output pixel (1 or 8)           */
    uint8_t down_mode;             /* KC_DOWN: 3 = k_lind (the only kernel of the class since round 4) */
    uint8_t out_wt;                /* output rows start on 64-byte boundaries: k_tile's stores go out write-through; else write-back (see fill_jobdev) */
    uint8_t dbg;                   /* -DSMOL_TUNING builds only (env SMOL_TILE_DBG): 1 no taps, 2 no staging, 4 no stores; else 0 */
    uint16_t tile_rows;            /* output rows per tile                               */
    uint16_t group_rows;           /* source rows staged per LDS phase                   */
    uint16_t launch_src_mb;        /* batches: source megabytes of ALL the batch's images (0: a call of its own); host side only, box2_setup */
    uint8_t box_force;             /* host side only, box2_setup: 64bpp-class box x box -- 0 decide by the launch's bytes, 1 single pass (k_tile), 2 two passes.
                                    * A strip that is cut into chunks decides ONCE and hands the verdict down */
    uint8_t hbox_force;            /* host side only, twopass_setup: bilinear x tall box (box2 == 3) -- 0 pick the splits of an output row's span from this
                                    * call's rows, n use n splits. A strip cut into chunks decides ONCE (its intermediate image is sized for the
                                    * strip's splits; a chunk of fewer rows would pick more: ADVICE round 5) */
    uint8_t tile_ring;             /* k_tile: the LDS holds a RING of group_rows source rows, refilled row by row (smol_kernel_tile.hip.h) */
    uint32_t span_chunks;          /* 16-byte chunks per staged row (LDS row pitch / 16) */
    uint32_t span_chunks_inv;      /* floor(2^32 / span_chunks) + 1                      */
    uint32_t box_nmax;             /* horizontal box: most interior pixels any column has */
    uint32_t span_px;              /* k_wtile: most source pixels a tile's span can hold  */
};

/* XCD-aware tile order (cdna_hip_programming.md T1). Workgroups are dealt round-robin to the chip's 8 XCDs, each
 * with its own L2: in raster order the 8 neighbours of a tile row run on 8 different L2s, and every 128-byte line
 * that two neighbouring spans share -- one per boundary and source row, 1/6 of a 768-byte span -- crosses the fabric
 * twice. With the swizzle each XCD takes a contiguous eighth of the tiles (of all jobs of the launch), in raster
 * order. 1-D launch of n workgroups; w -> (x, y, job) through host-made reciprocals (exact: the host only sets
 * `on' when n * gxgy < 2^32). */
struct TileSwz
{
    uint32_t on, gx, gxgy, inv_gx, inv_gxgy, q, r;
};

/* host: the swizzle for a (gx, gy, gz) grid, which then launches as n = gx * gy * gz workgroups in one dimension
 * (*n_out; 0 = keep the 3-D grid) */
inline TileSwz
swizzle_make (uint32_t gx, uint32_t gy, uint32_t gz, bool enable, uint32_t *n_out)
{
    TileSwz z = { 0, 0, 0, 0, 0, 0, 0 };
    const uint64_t gxgy = (uint64_t) gx * gy;
    const uint64_t n = gxgy * gz;

    *n_out = 0;
    if (!enable || n < 64 || n * gxgy >= ((uint64_t) 1 << 32))
        return z;
    z.on = 1;
    z.gx = gx;
    z.gxgy = (uint32_t) gxgy;
    /* floor (2^32 / d) + 1 does not fit 32 bits for d = 1: 0 stands for "divide by one" (a one-tile-wide image) */
    z.inv_gx = (gx == 1) ? 0u : (uint32_t) (((uint64_t) 1 << 32) / gx) + 1u;
    z.inv_gxgy = (gxgy == 1) ? 0u : (uint32_t) (((uint64_t) 1 << 32) / gxgy) + 1u;
    z.q = (uint32_t) (n / 8);
    z.r = (uint32_t) (n % 8);
    *n_out = (uint32_t) n;
    return z;
}

namespace dispatch {

/* byte position of R,G,B,A per SmolPixelType (smolscale.h:16-36 names the
 * memory order); 0xff = channel absent */
struct TypeInfo { uint8_t bpp; uint8_t pos [4]; uint8_t unassoc; };

static const TypeInfo k_types [10] = {
    { 4, { 0, 1, 2, 3 }, 0 }, { 4, { 2, 1, 0, 3 }, 0 }, { 4, { 1, 2, 3, 0 }, 0 }, { 4, { 3, 2, 1, 0 }, 0 },
    { 4, { 0, 1, 2, 3 }, 1 }, { 4, { 2, 1, 0, 3 }, 1 }, { 4, { 1, 2, 3, 0 }, 1 }, { 4, { 3, 2, 1, 0 }, 1 },
    { 3, { 0, 1, 2, 0xff }, 0 }, { 3, { 2, 1, 0, 0xff }, 0 },
};

enum KClass { KC_GENERAL = 0, KC_GENERAL_WIDE, KC_POW2, KC_TILE, KC_WTILE, KC_WAVEPX, KC_WAVEPX_WIDE, KC_POW2X, KC_UP, KC_DOWN, KC_BOX2, KC_HSPAN, KC_COUNT };

inline size_t align_up (size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr uint32_t TILE_LDS_BUDGET = 4 * 1024;      /* bytes of LDS per staging buffer (two per 64-lane workgroup) */

inline bool
tile_ok (const JobDev &d)
{
    return !d.wide;     /* 64bpp-class math: bilinear 0-2H, copy, one and box (ratio <= 255) */
}

/* 128bpp-class jobs without a box axis go to k_wtile */
inline bool
wtile_ok (const JobDev &d)
{
    if (!d.wide || d.filter_h == F_BOX || d.filter_v == F_BOX || tune_env ("SMOL_NO_WTILE"))
        return false;
    /* Measured (profiles/r01b_wide_kernels.log): converting each staged pixel once
     * pays when pixels or rows are re-used -- magnification (2.2-3x over k_general)
     * -- and loses ~25-60 % when every source pixel feeds a single tap. */
    const bool h_reuse = d.filter_h != F_BILINEAR || d.step_x < ((uint64_t) 1 << 32);
    const bool v_reuse = d.filter_v == F_ONE || (d.filter_v == F_BILINEAR && d.step_y < ((uint64_t) 1 << 32));
    return h_reuse || v_reuse;
}

/* Geometry of k_tile for this job: how many source pixels a 64-column tile can
 * span (bound from the closed-form sampling positions), how many source rows fit
 * one LDS phase, how many output rows a tile produces. */
inline void
tile_setup (JobDev &d)
{
    if (!tile_ok (d) && !wtile_ok (d))
        return;

    /* few taps per column (COPY/ONE/one sample) and a wide enough image: a lane
     * takes 4 adjacent columns, which amortises the wave's scalar bookkeeping and
     * turns its stores into 16-byte ones */
    d.out3_ok = (d.bpp_out == 3 && !d.unassoc_out && (!tune_env ("SMOL_OUT3_ALIGNED") || ((((uintptr_t) d.dst) | d.dst_stride) & 3) == 0));
    d.tile_cpl = (!d.wide && d.filter_h != F_BOX && d.filter_v != F_BOX
                  && (d.filter_h != F_BILINEAR || d.halv_h == 0) && d.w_out >= 256) ? 4 : 1;
    /* 24-bit output: 4 pixels per lane leave as one 12-byte store (measured: 4000x3000 ->
     * 1600x1200 RGB8 217 -> 168 us); worth the extra tap registers up to 2 samples per column */
    if (!d.wide && d.out3_ok && d.w_out >= 256 && d.filter_v != F_BOX
        && d.filter_h == F_BILINEAR && d.halv_h <= 1)
        d.tile_cpl = 4;
    /* ... and back to ONE column with row groups of 3 KB behind one halving on both axes where the 8 taps of four columns
     * stop sharing pixels or the source is premultiplied in LDS (4000 x 3000 -> 24-bit rows, profiles/r05_tile_cpl2.txt):
     * unassociated 32-bit sources at every ratio (2.1 169 -> 166 us, 2.5 152 -> 131, 3 150 -> 125, 3.8 150 -> 111),
     * premultiplied ones from ratio 3 (107 -> 103, 3.5 101 -> 93, 3.8 102 -> 92; at 2.1 four columns lead 123 : 138),
     * 24-bit ones from ratio 3.7 (3.5 equal, 3.8 104 -> 93, 3.99 95 -> 89) */
    bool one_col24 = !d.wide && d.tile_cpl == 4 && d.out3_ok && d.filter_h == F_BILINEAR && d.halv_h == 1
                           && d.filter_v == F_BILINEAR && d.halv_v == 1 && !tune_env ("SMOL_NO_TILE_CPL1_24")
                           && (d.bpp_in == 4 ? (d.unassoc_in ? ((((uintptr_t) d.src) | d.src_stride) & 3) == 0
                                                              : d.step_x >= ((uint64_t) 3 << 31))
                                             : d.step_x >= ((uint64_t) 185 << 32) / 100);
    if (one_col24)
        d.tile_cpl = 1;
    /* ... but not for a plain 32-bit premultiplied source shrunk by less than ~1.75 on this axis: once the
     * row loop's scalar work was cut (r02p), one column per lane is faster there (4096^2 -> 2731^2 123 -> 108 us,
     * -> 3413^2 174 -> 153, H down / V up 113 -> 82; at ratio 1.9, from 24-bit or unassociated sources 4 columns
     * stay 3-5 % ahead: profiles/r02p_cpl_ab.txt) */
    /* (round 3: that held only because the 4-column store demanded 16-byte aligned output rows, which those sizes
     * did not have. With vector stores on any dword-aligned row 4 columns win over the whole range 1-2 -- ratio 1.03
     * 107 -> 86 us, 1.2 130 -> 110, 1.5 98 -> 93, 1.7 96 -> 89 (profiles/r03d_cpl4_rows.txt) -- and one column per
     * lane stays for a vertical magnification under a horizontal minification and for destinations that are not
     * dword aligned) */
    if (d.tile_cpl == 4 && d.filter_h == F_BILINEAR && d.halv_h == 0 && d.bpp_in == 4 && !d.unassoc_in
        && ((((uintptr_t) d.src) | d.src_stride) & 3) == 0 && d.bpp_out == 4
        && d.step_x >= ((uint64_t) 1 << 32) && d.step_x < ((uint64_t) 7 << 30)
        && (tune_env ("SMOL_TILE_CPL1_BELOW_175") || (((uintptr_t) d.dst) | d.dst_stride) & 3
            || (d.filter_v == F_BILINEAR && d.step_y < ((uint64_t) 1 << 32))))
        d.tile_cpl = 1;
    /* one halving just above ratio 2 (the two samples of a column sit about a pixel apart, four columns share most of
     * their taps' pixels): 4 columns per lane -- 3840x2160 -> 1881x1057 168 -> 147 us; from ratio 2.5 on equal or behind
     * (profiles/r03d_cpl4_1h.txt) */
    if (!d.wide && d.tile_cpl == 1 && d.filter_h == F_BILINEAR && d.halv_h == 1 && d.filter_v == F_BILINEAR
        && d.bpp_in == 4 && !d.unassoc_in && ((((uintptr_t) d.src) | d.src_stride) & 3) == 0
        && d.bpp_out == 4 && ((((uintptr_t) d.dst) | d.dst_stride) & 3) == 0 && d.w_out >= 256
        && d.step_x < ((uint64_t) 115 << 32) / 100)                 /* per-sample step: ratio < 2.3 */
        d.tile_cpl = 4;
    /* 24-bit sources behind one horizontal halving into dword-aligned 32-bit rows (a photograph into an RGBA buffer): TWO
     * columns per lane and row groups of 3 KB -- the row loop's bookkeeping (42 of 118 us at ratio 2.5 by ablation:
     * profiles/r05_tile_cpl2.txt) is shared by twice the pixels, and the smaller buffers keep 26 instead of 20 waves on a
     * CU: 4000 x 3000 RGB8 -> RGBA ratio 2.1 156 -> 125 us, 2.5 125 -> 104, 3 96 -> 93, 3.8 92 -> 87. (32-bit sources:
     * ratio 2.5 equal, 3 -2 %, 3.5 and beyond +4 %, unassociated sources +8 %: they keep one column.) */
    bool two_cols = !d.wide && d.tile_cpl == 1 && d.bpp_in == 3 && d.filter_h == F_BILINEAR && d.halv_h == 1
                    && d.filter_v == F_BILINEAR && d.bpp_out == 4 && d.out_dword_ok && d.w_out >= 128
                    && !tune_env ("SMOL_NO_TILE_CPL2") && !tune_env ("SMOL_NO_TILE_VM");
    if (two_cols)
        d.tile_cpl = 2;
    if (const char *e = tune_env ("SMOL_TILE_CPL"))
        if (!d.wide && d.filter_h != F_BOX && d.filter_v != F_BOX)
        {
            two_cols = one_col24 = false;
            d.tile_cpl = (uint8_t) (atoi (e) == 4 ? 4 : 1);
            /* two columns per lane exist for bilinear x bilinear into dword-aligned 32-bit rows */
            if (atoi (e) == 2 && d.filter_h == F_BILINEAR && d.filter_v == F_BILINEAR && d.bpp_out == 4 && d.out_dword_ok
                && d.w_out >= 128 && !tune_env ("SMOL_NO_TILE_VM"))
                d.tile_cpl = 2;
        }
    uint64_t cols = 64u * d.tile_cpl;
    uint32_t box_lpc = 1;

    uint64_t span_px;
    if (d.filter_h == F_BOX)
    {
        /* narrow boxes: one lane per column; wide ones: 8 lanes per column, 8 columns per tile */
        uint64_t lpc8_min = 24;                 /* ratio above which a column gets 8 lanes */
        if (const char *e = tune_env ("SMOL_TILE_BOX_LPC8_MIN"))
            lpc8_min = (uint64_t) atoi (e);
        box_lpc = (d.step_x > (lpc8_min << 16)) ? 8 : 1;
        cols = 64u / box_lpc;
        span_px = ((cols * d.step_x) >> 16) + 2;
        d.box_nmax = (uint32_t) (d.step_x >> 16) + 1;
    }
    else if (d.filter_h == F_BILINEAR)
        span_px = (((cols << d.halv_h) * d.step_x) >> 32) + 3;
    else
        span_px = (d.filter_h == F_ONE) ? 1 : cols;
    if (span_px > d.w_in)
        span_px = d.w_in;

    /* up to 15 bytes of misalignment in front, 8 bytes of slack for the unaligned reader */
    const uint32_t bytes = (uint32_t) span_px * d.bpp_in + 15 + 8;
    d.span_chunks = (bytes + 15) / 16;
    d.span_chunks_inv = (uint32_t) (((uint64_t) 1 << 32) / d.span_chunks) + 1;

    d.span_px = (uint32_t) span_px;
    d.tile_single = 0;
    /* unassociated 32-bit sources that are premultiplied group by group in LDS (tapsel 4 below) take 3 KB too: 4096^2 rgbA ->
     * premultiplied ratio 1.1 100 -> 95 us, 2.1 181 -> 157, 3 137 -> 129, 3.9 125 -> 121, ratios 1.3-1.95 and 2.5 within 2 %
     * (profiles/r05_tile_cpl2.txt) */
    const bool lds_premul = !d.wide && d.bpp_in == 4 && ((((uintptr_t) d.src) | d.src_stride) & 3) == 0 && d.unassoc_in
                            && d.filter_h == F_BILINEAR && d.step_x >= ((uint64_t) 1 << 32)
                            && !tune_env ("SMOL_TILE_PREMUL_STAGE") && !tune_env ("SMOL_TILE_NO_LDS_PREMUL");
    uint32_t budget = (two_cols || lds_premul || one_col24) ? 3072u : TILE_LDS_BUDGET;
    if (const char *e = tune_env ("SMOL_TILE_LDS_BUDGET"))
        budget = (uint32_t) atoi (e);
#ifdef SMOL_TUNING
    if (const char *e = tune_env ("SMOL_TILE_DBG"))
        d.dbg = (uint8_t) atoi (e);
#endif
    uint32_t rows = budget / (d.span_chunks * 16);
    if (rows < 2 && 2 * d.span_chunks * 16 <= 8192)
        rows = 2;                   /* one-row groups leave the wave waiting on every row */
    if (rows < 1) rows = 1;
    if (rows > 16) rows = 16;
    d.group_rows = (uint16_t) rows;

    uint32_t th;
    if (d.filter_v == F_BOX)
    {
        /* about 64 source rows per tile */
        th = (uint32_t) (((uint64_t) 64 << 16) / d.step_y);
        if (th > 16) th = 16;
    }
    else if (d.filter_v == F_BILINEAR)
    {
        /* minification: 12-13 source rows per tile (profiles/r02p_tile_rows_sweep.txt: ratio 1.5 8-12 output
         * rows, ratio 2.5 and 3 four, ratio 6 two; below ~1.4 sixteen) */
        const uint64_t rows_per_out = d.step_y << d.halv_v;                 /* 32.32 source rows per output row */
        if (d.step_y < ((uint64_t) 1 << 32))
        {
            /* magnification on both axes (below x2.25: smol_jobdev.h up_setup sends it here): 12 rows
             * (3840x2160 x1.1: 12 rows 72 us, 8 rows 75, 32 rows 78; profiles/r03d_vec_store_align.txt) */
            th = (d.filter_h == F_BILINEAR && d.step_x < ((uint64_t) 1 << 32) && !d.wide) ? 12 : 32;
            /* (round 6) ... which was measured into ONE output set that the Infinity Cache kept. With the outputs really going to
             * memory (in rotation) and the arenas placed, flat tiles win for plain 32-bit pixels -- about 4.4 SOURCE rows per tile:
             * 3840x2160 x1.03 118 -> 99 us with 4 rows, x1.09 119 -> 106 (4), x1.17 134 -> 117 (5), x1.29 157 -> 132 (5),
             * x1.41 135 -> 112 (5-6), x1.65 149 -> 141 (6-8), x1.81 123 -> 112 (6), x2 and x2.15 within 2 % for 8-12
             * (profiles/r06_tile_rows_near1.txt) */
            if (th == 12 && d.bpp_in == 4 && !d.unassoc_in && d.bpp_out == 4 && d.out_dword_ok && !tune_env ("SMOL_TILE_ROWS_R5"))
            {
                th = (uint32_t) ((((uint64_t) 44 << 32) / 10 + d.step_y / 2) / d.step_y);
                if (th < 4) th = 4;
                if (th > 12) th = 12;
            }
        }
        else if (!d.wide && d.tile_cpl == 4 && d.halv_v == 0 && d.halv_h == 0 && d.filter_h == F_BILINEAR
                 && d.bpp_in == 4 && !d.unassoc_in && d.bpp_out == 4 && d.step_y < ((uint64_t) 2 << 32))
        {
            /* 4 columns per lane at ratios 1-2: about 8 source rows per tile (profiles/r03d_cpl4_rows.txt) */
            th = (uint32_t) ((((uint64_t) 8 << 32) + d.step_y / 2) / d.step_y);
            if (th < 4) th = 4;
            if (th > 8) th = 8;
            /* (round 6, outputs in rotation: at most 5 rows -- ratio 1.03 111 -> 107 us, 1.1 (3494 wide) 105 -> 98-100, 1.2 141 ->
             * 135, ratios 1.3-1.9 within 2 %; profiles/r06_tile_rows_near1.txt) */
            if (th > 5 && !tune_env ("SMOL_TILE_ROWS_R5"))
                th = 5;
        }
        else if (d.wide || (d.unassoc_in && d.bpp_in == 4))
            th = d.halv_v ? (16u >> d.halv_v) : 8u;                         /* k_wtile, and sources premultiplied group by group in LDS
                                                                             * (4000x3000 rgbA -> 1600x1200: 8 rows 137 us, 5 rows 146): as in round 1 */
        else if (d.tile_cpl == 4 && d.halv_h == 1 && d.halv_v == 1 && d.bpp_in == 4 && !d.unassoc_in && d.bpp_out == 4)
            th = 6;                                                         /* the ratio 2-2.3 family above */
        else if (rows_per_out < ((uint64_t) 14 << 32) / 10)                 /* < 1.4 */
            th = 16;
        else
        {
            th = (uint32_t) (((uint64_t) 13 << 32) / rows_per_out);
            if (th < 2) th = 2;
        }
    }
    else
        th = (d.filter_v == F_ONE) ? 32 : 16;
    if (const char *e = tune_env ("SMOL_TILE_ROWS"))
        th = (uint32_t) atoi (e);
    if (th > (TILE_MAX_VSAMPLES >> (d.filter_v == F_BILINEAR ? d.halv_v : 0)))
        th = TILE_MAX_VSAMPLES >> (d.filter_v == F_BILINEAR ? d.halv_v : 0);    /* the schedule is one register: lane k = sample k, lane 63 stays a sentinel */
    if (th < 1)
        th = 1;
    d.tile_rows = (uint16_t) th;
    d.tile_aligned4 = (d.bpp_in == 4 && ((((uintptr_t) d.src) | d.src_stride) & 3) == 0);

    uint32_t tapsel;
    if (d.bpp_in == 3)
        tapsel = 3;
    else if (!d.tile_aligned4)
        tapsel = 2;
    else if (d.unassoc_in)
    {
        /* premultiply once per staged pixel when pixels are re-used (magnification);
         * when every pixel feeds one tap, premultiply at the tap and keep the
         * double-buffered DMA staging */
        const bool reuse = d.filter_h != F_BILINEAR || d.step_x < ((uint64_t) 1 << 32)
                           || tune_env ("SMOL_TILE_PREMUL_STAGE");
        tapsel = reuse ? 1 : (tune_env ("SMOL_TILE_NO_LDS_PREMUL") ? 2 : 4);   /* 4: DMA, then premultiply the group in LDS */
    }
    else
        tapsel = 0;

    uint32_t hm = (d.filter_h == F_BILINEAR) ? 1u + d.halv_h : 0u;
    if (d.filter_h == F_BOX)
        hm = (box_lpc == 8) ? 5 : 4;
    d.tile_lpc = (uint8_t) box_lpc;

    uint32_t out;
    if (d.bpp_out == 4 && d.out_dword_ok)
        out = d.unassoc_out ? 1 : 0;
    else
        out = 2;

    d.tile_variant = (uint8_t) ((tapsel << 5) | (hm << 2) | out);
    d.tile_band = ((d.w_out + cols - 1) / cols >= 64 && !tune_env ("SMOL_NO_TILE_BAND"));
    /* vertical mode compiled in for the combinations that carry the traffic (the rest decide at run time):
     * bilinear H with copy / bilinear V, box H with box V */
    d.tile_vm = TILE_VM_ANY;
    if (d.filter_v == F_COPY && d.tile_cpl == 1 && hm >= 1 && hm <= 3 && !tune_env ("SMOL_NO_TILE_VCOPY"))
        d.tile_vm = TILE_VM_COPY;
    else if (d.filter_v == F_BILINEAR && hm >= 1 && hm <= 3 && !tune_env ("SMOL_NO_TILE_VM"))
        d.tile_vm = TILE_VM_PAIR;
    else if (d.filter_v == F_BOX && hm >= 4 && !tune_env ("SMOL_NO_TILE_VM"))
        d.tile_vm = TILE_VM_BOX;
    /* 4-byte alignment is all a 16-byte store needs (global_store_dwordx4 wants dword alignment; the lanes' stores are
     * contiguous either way). Rounds 1-2 demanded 16-byte aligned rows and sent every odd output width -- three sizes
     * out of four in the reference's proportional sweep -- to per-pixel stores: 1080p -> 4993x2809 (k_up) 206 -> 97 us,
     * 4K -> 5261x2959 185 -> 115, ratio 1.03 with 4 columns per lane 139 -> 89 (profiles/r03d_vec_store_align.txt).
     * SMOL_VEC16_ALIGNED (tuning builds): the old rule, for A/B. */
    d.out_vec16_ok = (d.bpp_out == 4 && ((((uintptr_t) d.dst) | d.dst_stride) & (tune_env ("SMOL_VEC16_ALIGNED") ? 15 : 3)) == 0);
    /* (any row alignment since round 3: the dword-granular 24-bit store shifts the tile's byte run onto aligned dwords
     * itself, smol_kernel_tile.hip.h tile_store_cols; SMOL_OUT3_ALIGNED in tuning builds = the old 4-byte rule, for A/B) */
    d.out3_ok = (d.bpp_out == 3 && !d.unassoc_out && (!tune_env ("SMOL_OUT3_ALIGNED") || ((((uintptr_t) d.dst) | d.dst_stride) & 3) == 0));

    /* Ring staging (round 6; smol_kernel_tile.hip.h): the LDS holds a ring of source rows that is refilled row by row, so that
     * N DMA instructions stay in flight behind the row being consumed -- instead of two buffers of which one is in flight and a
     * tile that pays the memory latency once per row group. Plain DMA staging only (tapsel 0, 2, 3), bilinear on both axes,
     * both minifying or both magnifying, N = 4: the reference's proportional sweep (every 5th size, device resident) one
     * kernel per image 1080p 7.17e11 -> 7.47e11 pixels per second, 4K 1.03e12 -> 1.08e12, 8K 1.11e12 -> 1.16e12; as one launch
     * 1.335e12 -> 1.369e12, 4K and 8K within the runs' spread; 4096^2 -> 1365^2 115 -> 109 us, -> 1170^2 107 -> 103, -> 819^2
     * 107 -> 104, -> 2731^2 96 -> 93.5, 3840x2160 x1.1 73.7 -> 71.4. Deeper rings lose again (N = 12: every row slower), a box axis
     * and mixed axes lose with every depth (4096x2048 -> 2731x3000 75 -> 89 us, 4096^2 -> 256^2 98 -> 106): they keep the two
     * buffers (profiles/r06_tile_ring.txt). The ring holds enough rows that N instructions lie behind any row of an interior
     * tile: such a row takes at least kmin instructions of up to 1 KB. */
    d.tile_ring = 0;
    if (!d.wide && (tapsel == 0 || tapsel == 2 || tapsel == 3) && !tune_env ("SMOL_NO_TILE_RING"))
    {
        uint32_t n = 4;
        if (const char *e = tune_env ("SMOL_TILE_RING"))
            n = (uint32_t) atoi (e);
        /* which jobs: 1 bilinear minification on both axes, 2 magnification on both, 4 one of each, 8 a box axis, 16 a COPY / ONE axis */
        const bool bil = d.filter_h == F_BILINEAR && d.filter_v == F_BILINEAR;
        const bool dn_x = d.step_x >= ((uint64_t) 1 << 32), dn_y = d.step_y >= ((uint64_t) 1 << 32);
        const uint32_t cls = (d.filter_h == F_BOX || d.filter_v == F_BOX) ? 8u : !bil ? 16u : (dn_x && dn_y) ? 1u : (!dn_x && !dn_y) ? 2u : 4u;
        const uint32_t want = tune_env ("SMOL_TILE_RING_CLASS") ? (uint32_t) atoi (tune_env ("SMOL_TILE_RING_CLASS")) : 3u;
        if (!(cls & want))
            n = 0;
        if (n == 2 || n == 3 || n == 4 || n == 6 || n == 8 || n == 12)
        {
            uint64_t px_min;
            if (d.filter_h == F_BOX)
                px_min = (cols * d.step_x) >> 16;
            else if (d.filter_h == F_BILINEAR)
                px_min = ((((cols << d.halv_h) - 1u) * d.step_x) >> 32) + 1u;
            else
                px_min = (d.filter_h == F_ONE) ? 1 : cols;
            if (px_min > d.w_in)
                px_min = d.w_in;
            const uint32_t kmin = ((((uint32_t) px_min * d.bpp_in + 15u) & ~15u) + 1023u) / 1024u;
            uint32_t ring_rows = (n + kmin - 1u) / kmin + 1u;
            if (const char *e = tune_env ("SMOL_TILE_RING_ROWS"))
                ring_rows = (uint32_t) atoi (e);
            if (ring_rows >= 2u && ring_rows <= 64u && ring_rows * d.span_chunks * 16u <= 48u * 1024u)
            {
                d.tile_ring = (uint8_t) n;
                d.group_rows = (uint16_t) ring_rows;
                d.tile_single = 1;              /* (one region of group_rows rows: tile_lds_bytes) */
                return;
            }
        }
    }

    /* ONE staging buffer of twice the size and no prefetch of the next row group (the resident waves cover
     * each other's latency, as in k_boxh) for the entry points where it measured ahead -- compiled in there
     * (tile_single_variant() in smol_kernel_tile.hip.h states the same condition): unassociated sources
     * premultiplied in LDS, whose pass likes long groups (4000x3000 rgbA 138 -> 132 us), and 32-bit sources at one
     * sample and one column per lane under a bilinear vertical filter (4096^2 -> 2731^2 112 -> 106). Elsewhere
     * equal or behind (box 108 -> 113; profiles/r02w_tile_single.txt). */
    if (!d.wide && !tune_env ("SMOL_TILE_NO_SINGLE")
        && (tapsel == 4 || (tapsel == 0 && hm == 1 && d.tile_cpl == 1 && d.tile_vm == TILE_VM_PAIR)))
    {
        d.tile_single = 1;
        uint32_t rows2 = 2 * budget / (d.span_chunks * 16);
        if (rows2 < 1) rows2 = 1;
        if (rows2 > 16) rows2 = 16;
        if (rows2 > d.group_rows)
            d.group_rows = (uint16_t) rows2;
    }
}

inline uint32_t
tile_lds_bytes (const JobDev &d)
{
    uint32_t bytes = (d.tile_single ? 1u : 2u) * d.group_rows * d.span_chunks * 16 + 32;
    if (!d.wide && d.unassoc_out)
        bytes += 1024;                          /* k_tile: LDS copy of the inverse-alpha table */
    if (d.wide)
        bytes += d.group_rows * d.span_px * ((d.p16 && d.linear) ? 16 : 8) + 16   /* converted rows (k_wtile) */
                 + SMOL_LUT_LDS_BYTES;                                            /* + LDS copy of the tables */
    return bytes;
}

inline void
fill_jobdev (JobDev &d, const SmolHipJob &j, const void *src, uint32_t src_stride,
             void *dst, uint32_t dst_stride, uint32_t first_row, uint32_t n_rows)
{
    const TypeInfo &ti = k_types [j.pixel_type_in];
    const TypeInfo &to = k_types [j.pixel_type_out];

    memset (&d, 0, sizeof (d));
    d.src = static_cast<const uint8_t *> (src);
    d.dst = static_cast<uint8_t *> (dst);
    d.src_stride = src_stride;
    d.dst_stride = dst_stride;
    d.w_in = j.width_in;  d.h_in = j.height_in;
    d.w_out = j.width_out; d.h_out = j.height_out;
    d.first_row = first_row; d.n_rows = n_rows;
    d.frac0_x = j.frac0_x; d.step_x = j.step_x;
    d.frac0_y = j.frac0_y; d.step_y = j.step_y;
    d.span_mul_x = j.span_mul_x; d.span_mul_y = j.span_mul_y;
    d.filter_h = j.filter_h; d.filter_v = j.filter_v;
    d.halv_h = j.halvings_h; d.halv_v = j.halvings_v;
    d.bpp_in = ti.bpp; d.bpp_out = to.bpp;
    for (int i = 0; i < 4; i++)
    {
        d.pos_in [i] = ti.pos [i];
        d.pos_out [i] = to.pos [i];
    }
    /* R and B are labels: nothing between unpack and pack tells the colour channels apart (every conversion and every
     * filter treats them alike), so the two are swapped on BOTH sides whenever the source holds B in front of R. The
     * kernels then only ever see colour bytes in ascending memory order -- 0,1,2 (alpha absent or last) or 1,2,3 (alpha
     * first): k_lind's three-channel flavour reads bytes 0,1,2 with compile-time selectors (one SDWA instruction where a
     * run-time position costs a v_bfe and a shift). (The same in WideUnpack for 32-bit sources -- SDWA shifts and multiplies,
     * alpha-first words rotated -- measured neutral, -3 ... +4 % by row, and was not kept: profiles/r05_lind3_ab.txt.) The
     * 64bpp class works on whole pixel words and out_perm (below), which is made from the types themselves. */
    if (d.pos_in [0] > d.pos_in [2])
    {
        const uint8_t ti0 = d.pos_in [0], to0 = d.pos_out [0];
        d.pos_in [0] = d.pos_in [2]; d.pos_in [2] = ti0;
        d.pos_out [0] = d.pos_out [2]; d.pos_out [2] = to0;
    }
    d.unassoc_in = ti.unassoc; d.unassoc_out = to.unassoc;
    d.wide = j.wide; d.p16 = j.premul16; d.linear = j.linear; d.quirk24 = j.quirk24;
    d.in_dword_ok = (ti.bpp == 4 && (((uintptr_t) src | src_stride) & 3) == 0);
    d.out_dword_ok = (to.bpp == 4 && (((uintptr_t) dst | dst_stride) & 3) == 0);
    /* Write-through stores keep the output stream out of the L2 -- worth 1-2 % when a wave's kilobyte of output is whole
     * 128-byte lines. When rows start anywhere (every output width that is not a multiple of 16 pixels: nearly every size
     * of the reference's sweep) the first and last line of each kilobyte are partial, a write-through of a partial line is
     * a read-modify-write in memory, and the neighbouring tile does the same to the other part later: 3840x2160 ->
     * 4185x2354 180 us write-through, 156 us write-back; -> 3724x2094 253 / 221 (profiles/r04_store_policy_ab.txt).
     * Write-back lets the two parts meet in the L2. */
    d.out_wt = ((((uintptr_t) dst | dst_stride) & 63) == 0 && !tune_env ("SMOL_TILE_NO_WT")) || tune_env ("SMOL_TILE_WT");

    /* byte shuffle from the internal pixel word (bytes in INPUT memory order; a
     * 24bpp input gets alpha = 0xff as byte 3) to the output memory order */
    {
        uint32_t sel = 0;
        for (int ch = 0; ch < 4; ch++)
        {
            uint32_t from = (ch == 3 && ti.bpp == 3) ? 3u : ti.pos [ch];
            if (to.pos [ch] != 0xff)
                sel |= from << (8 * to.pos [ch]);
        }
        d.out_perm = sel;
    }

    tile_setup (d);
}

/* Does the job qualify for k_pow2? Exact 2:1/4:1/8:1 on both axes, 32-bit
 * premultiplied in and out, 64bpp-class math, 16-byte aligned source rows. */
inline bool
pow2_ok (const JobDev &d)
{
    if (d.filter_h != F_BILINEAR || d.filter_v != F_BILINEAR)
        return false;
    if (d.wide || d.bpp_in != 4 || d.bpp_out != 4 || d.unassoc_in || d.unassoc_out)
        return false;
    if (d.w_in != (d.w_out << (d.halv_h + 1)) || d.h_in != (d.h_out << (d.halv_v + 1)))
        return false;

    const uint32_t rx = 2u << d.halv_h;
    const uint32_t opt = rx >= 4 ? 1 : 4 / rx;

    if (d.w_out % opt)
        return false;
    /* 4-byte aligned rows suffice: global_load_dwordx4 only needs dword alignment, and a sub-rectangle of a
     * larger image (pixels_in + 4 * x, the parent's stride: smolscale.h:56-60) keeps the streaming kernel.
     * SMOL_POW2_ALIGN16 (tuning builds): the old 16-byte requirement, for A/B */
    if ((((uintptr_t) d.src) | d.src_stride) & (tune_env ("SMOL_POW2_ALIGN16") ? 15 : 3))
        return false;
    if ((((uintptr_t) d.dst) | d.dst_stride) & (opt * 4 - 1))
        return false;
    return true;
}

/* k_pow2 workgroup height: output rows (= waves) per workgroup */
/* (rows that share 128-byte lines with the neighbouring column tiles -- crops, odd widths: smol_k_stream.hip -- run
 * 8 rows per workgroup under the XCD-aware order: 4096^2 -> 1024^2 at base + 4 B / parent stride 97 -> 91 us, aligned
 * rows 85 -> 88: profiles/r03a_pow2_crop.txt) */
inline bool
pow2_shared_lines (const JobDev &d)
{
    return ((((uintptr_t) d.src) | d.src_stride) & 255u) != 0;
}

inline uint32_t
pow2_block_rows (const JobDev &d)
{
    uint32_t by = pow2_shared_lines (d) ? 8 : 16;
    /* (round 6, outputs in rotation: four rows per workgroup for exact 2:1 of small frames looked 4 % ahead on BASELINE config 1 in a
     * knob sweep and 0-3 % in an interleaved A/B, behind by 5 % on 4096^2 -> 2048^2: not kept; profiles/r06_retune_rot.txt) */
    if (const char *e = tune_env ("SMOL_POW2_BY"))
        if (atoi (e) >= 1 && atoi (e) <= 16)
            by = (uint32_t) atoi (e);
    return by;
}

/* k_pow2x geometry: output pixels per lane for this job's kind and ratio */
inline uint32_t
pow2x_opt (const JobDev &d)
{
    const uint32_t r = 2u << d.halv_h;
    return (r >= 4 ? r : 4u) / r;          /* a lane's load unit is 4 source pixels */
}

/* Exact 2:1/4:1/8:1 with the same ratio on both axes, any pixel types, either
 * storage class; 16-byte aligned source rows, destination aligned for the lane's
 * vector store. Fills px_kind / px_lpr_shift. */
inline bool
pow2x_setup (JobDev &d)
{
    if (d.filter_h != F_BILINEAR || d.filter_v != F_BILINEAR || d.halv_h != d.halv_v)
        return false;
    if (d.w_in != (d.w_out << (d.halv_h + 1)) || d.h_in != (d.h_out << (d.halv_v + 1)))
        return false;
    if (tune_env ("SMOL_NO_POW2X"))
        return false;

    const uint32_t opt = pow2x_opt (d);

    if (d.w_out % opt)
        return false;
    if ((((uintptr_t) d.src) | d.src_stride) & ((d.bpp_in == 3 || !tune_env ("SMOL_POW2_ALIGN16")) ? 3 : 15))
        return false;
    if (!d.wide)
    {
        const uint32_t dalign = (d.bpp_out == 4) ? (opt >= 4 ? 16u : opt * 4u)
                                                 : (opt >= 4 ? 4u : (opt == 2 ? 2u : 1u));
        if ((((uintptr_t) d.dst) | d.dst_stride) & (dalign - 1))
            return false;
    }

    d.px_kind = (uint8_t) ((d.wide ? 2 : 0) + (d.bpp_in == 3 ? 1 : 0));
    if (d.wide && d.bpp_in == 4 && d.p16 && !d.linear)
        d.px_kind = PX_W16;
    const uint32_t tcols = d.w_out / opt;
    uint32_t shift = 0;
    while (shift < 6 && (1u << shift) < tcols)
        shift++;
    d.px_lpr_shift = (uint8_t) shift;
    return true;
}

/* k_pow2x grid: (column tiles, row tiles) -- a 2-D grid, so no division at the head of a workgroup */
inline uint32_t
pow2x_tiles_x (const JobDev &d)
{
    const uint32_t tcols = d.w_out / pow2x_opt (d);
    const uint32_t lpr = 1u << d.px_lpr_shift;
    return (tcols + lpr - 1) / lpr;
}

inline uint32_t
pow2x_tiles_y (const JobDev &d)
{
    const uint32_t rows_per_block = 16u * (64u >> d.px_lpr_shift);
    return (d.n_rows + rows_per_block - 1) / rows_per_block;
}

/* k_up: bilinear magnification on both axes, 64bpp class. A tile of tile_rows
 * output rows may touch at most UP_NSRC source rows (smol_kernel_up.hip.h):
 * rows r = 0..TR-1 span floor ((TR-1) * step) + 3 source rows at most. */
constexpr uint32_t DOWN_LUT_BYTES_FWD = (SMOL_LUT_LDS_BYTES + 15) & ~15u;   /* LDS copy of the tables */
constexpr uint32_t UP_NSRC_HOST = 10;
constexpr uint32_t UP_NSRC_W_HOST = 8;      /* premul16 flavour */
#ifndef SMOL_UPG_NSRC
#define SMOL_UPG_NSRC 8
#endif
constexpr uint32_t UPG_NSRC_HOST = SMOL_UPG_NSRC;      /* table-driven flavours (k_upg): source rows a tile keeps in registers
                                                        * (6 or 5 rows -- fewer registers, flatter tiles -- measured within +-3 % and
                                                        * -8 %: profiles/r03l_upg_nsrc_ab.txt) */

inline bool
up_setup (JobDev &d)
{
    /* flavour: 0 64bpp class; 1 premul16 without sRGB on 32-bit pixels; 2 other 128bpp-class
     * jobs with 16-bit channels (sRGB-linear, 8-bit alpha); 3 premul16 + sRGB (19-bit channels) */
    uint32_t flavour = 0;
    if (d.wide)
        flavour = (d.p16 && !d.linear && d.bpp_in == 4 && d.bpp_out == 4) ? 1u : ((d.p16 && d.linear) ? 3u : 2u);

    if (d.filter_h != F_BILINEAR || d.filter_v != F_BILINEAR || d.halv_h || d.halv_v)
        return false;
    if (d.step_x >= ((uint64_t) 1 << 32) || d.step_y >= ((uint64_t) 1 << 32) || d.step_x == 0 || d.step_y == 0)
        return false;
    if (d.w_in < 2 || d.h_in < 2 || tune_env ("SMOL_NO_UP") || (flavour >= 2 && tune_env ("SMOL_NO_UP_GEN"))
        || (flavour == 1 && tune_env ("SMOL_NO_UP_P16")))
        return false;

    /* 64bpp class, slight magnification (below 2 on either axis): every source row feeds barely more than one
     * output row, k_up's per-tile horizontal pass over 10 source rows is not amortised and the row-streaming k_tile
     * wins -- 3840x2160 x1.1: 143 -> 79 us, x1.3: 154 -> 105, x1.5: 156 -> 133, x1.75: equal, x3: k_up 104 against
     * 142 (profiles/r03d_up_vs_tile.txt). This is the upper half of the reference's own proportional sweep. */
    if (flavour == 0 && !tune_env ("SMOL_UP_ALWAYS"))
    {
        /* (between 1.7 and 2 the two are equal on 16-byte aligned output rows and k_tile is 6-12 % ahead on the odd
         * widths of the sweep: 3840x2160 -> 6758x3801 641 -> 569 us, -> 7411x4168 674 -> 634: profiles/r03d_sweep_points.txt) */
        /* (round 5, kernel against kernel in steps of 0.1 on aligned and odd output widths, profiles/r05_up_handover.txt:
         * x2.0 k_tile 65 / 68 us against k_up 69 / 82, x2.2 72 / 74 against 76 / 83, x2.3 82 / 93 against 80 / 88: k_up
         * from x2.25) */
        const uint64_t lim = ((uint64_t) 4 << 32) / 9;              /* step 0.444 = magnification 2.25 */
        if (d.step_x > lim || d.step_y > lim)
            return false;
        /* k_up's fast form is the 16-byte store of four premultiplied 32-bit pixels; 24-bit, unassociated and byte-aligned
         * output leaves it through k_tile's column stores and loses to k_tile at every factor: 1920x1080 -> x4, 4 frames,
         * RGB8 -> RGB8 179 / 120 us, RGBA -> RGB8 154 / 113, premultiplied -> unassociated 230 / 153; 3840x2160 -> x2, 6 frames,
         * 401 / 185, 287 / 184, 561 / 241; x8 151 / 101, 178 / 125 (profiles/r04_up_vs_lind.txt). 24-bit sources into
         * premultiplied 32-bit: k_up from x2.6 (x2 243 / 181, x2.6 118 / 117, x3 116 / 141, x4 91 / 99) */
        /* (round 5: 24-bit output leaves k_up as the lane's twelve bytes -- an 8-byte and a 4-byte streaming store:
         * RGBA -> RGB8 x2.25 equal, x2.4 odd 58 -> 49, x2.6 66 -> 55 us, x3 68 -> 62, x4 111 -> 86, x8 100 -> 75, unassociated -> BGR8 x4 on an odd
         * width 158 -> 100; RGB8 -> RGB8 x2.6 67 -> 65, x3 73 -> 63, x4 110 -> 87, x4 odd 140 -> 100, x8 100 -> 75:
         * profiles/r05_up_rgb8.txt) */
        const bool out24 = d.out3_ok && !tune_env ("SMOL_NO_UP_P3");
        /* (... and unassociated 32-bit output un-premultiplied in registers through an LDS copy of the inverse table, in the
         * 16-byte store form, from x3.2: the multiplies make it arithmetic-bound -- premultiplied -> unassociated x2.25 58 ->
         * 67 us and x2.6 77 -> 83 (k_tile stays), x3 equal, x3.3 108 -> 92, x4 154 -> 115, x4 on an odd width 166 -> 127,
         * x8 129 -> 105: profiles/r05_up_rgb8.txt) */
        const bool out_u4 = (d.tile_variant & 3u) == 1u /* OUT_U4 */ && d.out_vec16_ok && !tune_env ("SMOL_NO_UP_U4");
        if ((d.tile_variant & 3u) != 0 && !out24 && !out_u4)
            return false;
        if (out_u4 && !tune_env ("SMOL_UP_ALWAYS") && (d.step_x > ((uint64_t) 5 << 32) / 16 || d.step_y > ((uint64_t) 5 << 32) / 16))
            return false;
        /* (24-bit sources keep their own handover at x2.6, below; 32-bit sources into 24-bit rows take k_up from x2.25 like
         * everything else: x2.25 on an aligned width equal (46 us), x2.4 on an odd one 58 -> 49) */
        if (d.bpp_in == 3 && (d.step_x > ((uint64_t) 5 << 32) / 13 || d.step_y > ((uint64_t) 5 << 32) / 13))
            return false;
    }

    /* 128bpp class: k_up's tiles keep the horizontal results of 8 source rows in registers and pay for them per tile -- worth it
     * when every source row feeds several output rows. Below that the row-streaming k_lind (down_setup takes what is
     * refused here) is ahead, as k_tile is in the 64bpp class; step us on 8-12 frames, k_up / k_upg against k_lind
     * (profiles/r04_up_vs_lind.txt):
     *   gamma-correct RGBA      x1.1 435 / 156   x1.5 403 / 245   x2 416 / 368   x2.6 210 / 202   x3 250 / 237   x4 191 / 201
     *   premul16 + gamma        x1.1 341 / 168   x1.5 350 / 261   x2 470 / 389   x2.6 198 / 218   x3 214 / 253   x4 169 / 215
     *   gamma-correct RGB8 (24-bit out: k_upg stores pixel by pixel)    x1.1 490 / 151   x2 504 / 347   x4 269 / 184
     *   premul16                x1.1 158 / 149   x1.5 182 / 234   x2 227 / 354   x4 112 / 197 */
    if (flavour != 0 && !tune_env ("SMOL_UP_ALWAYS"))
    {
        const uint64_t one = (uint64_t) 1 << 32;
        const uint64_t smax = d.step_x > d.step_y ? d.step_x : d.step_y;         /* the less magnified axis */
        const uint64_t lim = flavour == 1 ? one * 5 / 6                            /* premul16: k_up from x1.2 */
                           : flavour == 3 ? one * 10 / 23                          /* premul16 + gamma: k_upg from x2.3 */
                           : one * 2 / 7;                                          /* gamma-correct: k_upg from x3.5 */
        if (smax > lim || (flavour >= 2 && d.bpp_out == 3))
            return false;
    }

    d.up_wide = (uint8_t) flavour;
    d.tile_band = tune_env ("SMOL_NO_UP_BAND") ? 0 : 1;     /* tiles dealt out in XCD bands */
    uint64_t tr = (((uint64_t) ((flavour >= 2 ? UPG_NSRC_HOST : flavour ? UP_NSRC_W_HOST : UP_NSRC_HOST) - 3)) << 32) / d.step_y + 1;
    if (tr > 64)
        tr = 64;
    /* 64bpp class: the kernel is bound by its stores, and those want the chip to sweep the output in
     * (nearly) raster order -- flat tiles. Config 3: 8 rows 89-92 us, 7 or 9 rows 92, 6 rows 98, 12-29
     * rows 97-104, 4 rows 124 (per-tile set-up); profiles/r02e_up_tile_rows.txt. The table-driven
     * flavours are bound by arithmetic per tile and keep the tall tiles. */
    /* (... below x2.75 twelve: a tile's two halo rows of horizontal work are shared by more output rows -- x2.4 83 -> 77 us,
     * x2.6 91 -> 88, odd widths 95 -> 87 and 99 -> 97; from x2.8 eight rows are ahead, 95 against 98: profiles/r05_up_handover.txt) */
    /* (unassociated output is bound by its multiplies, not by its stores, and copies the inverse table into LDS once per tile:
     * twenty rows -- x4 128 -> 115 us, on an odd width 138 -> 127, x8 118 -> 105: profiles/r05_up_rgb8.txt) */
    /* (round 6, outputs in rotation: 24-bit rows from x2.75 sixteen -- RGB8 -> RGB8 x4 87 -> 74 us, RGBA -> RGB8 x4 85 -> 70, x3 89 -> 85;
     * from x6 twelve, x8 77 -> 70; below x2.75 twelve as before; profiles/r06_retune_rot.txt) */
    const bool out24_rows = flavour == 0 && d.out3_ok && (d.tile_variant & 3u) == 2u /* OUT_ANY */ && !tune_env ("SMOL_UP_ROWS_R5");
    const uint64_t flat_rows = (flavour == 0 && (d.tile_variant & 3u) == 1u /* OUT_U4 */) ? 20
                             : (flavour == 0 && d.step_y > ((uint64_t) 4 << 32) / 11) ? 12
                             : out24_rows ? (d.step_y > ((uint64_t) 1 << 32) / 6 ? 16 : 12) : 8;
    if (flavour == 0 && tr > flat_rows)
        tr = flat_rows;
    if (flavour != 0 && tr > 16)
        tr = 16;                /* 1080p -> 4320p: sRGB-linear 224 -> 203 us, premul16 70 -> 67.5 (profiles/r02o_up_tile_rows.txt) */
    if (const char *e = tune_env ("SMOL_UP_ROWS"))
    {
        /* (up to what the tile's source rows allow) */
        const uint64_t lim = (((uint64_t) ((flavour >= 2 ? UPG_NSRC_HOST : flavour ? UP_NSRC_W_HOST : UP_NSRC_HOST) - 3)) << 32) / d.step_y + 1;
        if ((uint64_t) atoi (e) >= 1)
            tr = (uint64_t) atoi (e) < lim ? (uint64_t) atoi (e) : lim;
        if (tr > 64)
            tr = 64;
    }
    d.tile_rows = (uint16_t) tr;

    /* source pixels a tile can reach: ofs (x0 + cols - 1) + 1 - ofs (x0) + 1 */
    const uint32_t cols = (flavour == 3) ? 128u : 256u;
    uint64_t span = (((cols - 1u) * d.step_x) >> 32) + 3;
    if (span > d.w_in)
        span = d.w_in;
    d.span_px = (uint32_t) span;
    d.span_chunks_inv = (uint32_t) (((uint64_t) 1 << 32) / span) + 1;     /* i / span for i < 2^16 */

    /* the four result bytes sit in the high byte of each 16-bit lane of {hi, lo}:
     * in-order byte k is v_perm index {1, 5, 3, 7}[k]; compose with out_perm */
    static const uint32_t gather [4] = { 1, 5, 3, 7 };
    d.up_perm = 0;
    for (int i = 0; i < 4; i++)
        d.up_perm |= gather [(d.out_perm >> (8 * i)) & 3u] << (8 * i);
    return true;
}

inline uint32_t
up_lds_bytes (const JobDev &d)
{
    if (d.up_wide >= 2)
        return DOWN_LUT_BYTES_FWD + 4u * ((UPG_NSRC_HOST * d.span_px * (d.up_wide == 3 ? 16u : 8u) + 15u) & ~15u) + 16u;
    if (d.up_wide)
        return UP_NSRC_W_HOST * d.span_px * 8u + 1024u + 16u;       /* + the inverse table */
    /* (unassociated output: + the inverse table behind the staged rows) */
    return ((UP_NSRC_HOST * d.span_px * 8u + 15u) & ~15u) + (d.unassoc_out ? 1024u : 0u) + 16u;
}

/* tiles of the job, rounded up to a multiple of 8 (flavours 2-3 launch one workgroup per 4 tiles) */
inline uint32_t
up_blocks (const JobDev &d)
{
    const uint32_t tw = (d.up_wide == 3) ? 128u : 256u;
    const uint32_t n = ((d.w_out + tw - 1u) / tw) * ((d.n_rows + d.tile_rows - 1u) / d.tile_rows);
    return (n + 7u) & ~7u;
}

/* KC_DOWN: bilinear minification on both axes (0-2 halvings) for the 128bpp class = k_lind (smol_kernel_lind.hip.h):
 * premul16 with and without sRGB, the sRGB-linear flavours, 32- and 24-bit rows at any byte alignment. 4 waves per
 * workgroup around one LDS copy of the tables. */
constexpr uint32_t DOWN_LUT_BYTES = (SMOL_LUT_LDS_BYTES + 15) & ~15u;

/* k_lind's tile width W <= 128 (in group_rows): chosen so that the span of source pixels a tile reaches just fills
 * whole conversion rounds of 64 pixels -- K rounds hold 64 K pixels, a tile of W columns reaches at most
 * ((W << h) - 1) * step + 3 of them. */
constexpr uint32_t LIND_KMAX_HOST = 5;
constexpr uint32_t LIND_KMAX_NARROW_HOST = 8;       /* the fused kernel's one-column flavour (one or two halvings, tiles of <= 64 columns) */

inline uint64_t
lind_span (const JobDev &d, uint32_t w)
{
    return ((((uint64_t) (w << d.halv_h) - 1u) * d.step_x) >> 32) + 3u;
}

inline bool lind_narrow (const JobDev &d) { return d.halv_h >= 1 && d.group_rows <= 64; }

/* k_lind with streaming stores (a kernel of its own: smol_kernel_lind.hip.h, lind3_body): bilinear magnification on both axes
 * into rows that start on 64-byte boundaries */
inline bool
lind_nt (const JobDev &d)
{
    return d.out_wt && d.halv_h == 0 && d.filter_h == F_BILINEAR && d.filter_v == F_BILINEAR
           && d.step_x < ((uint64_t) 1 << 32) && d.step_y < ((uint64_t) 1 << 32) && !tune_env ("SMOL_NO_LIND_NT");
}

/* k_lind with the vertical sample count compiled in: no halving on either axis (the streaming flavour above has it too) */
inline bool lind_hv0 (const JobDev &d) { return d.halv_h == 0 && d.halv_v == 0 && !tune_env ("SMOL_NO_LIND_HV0"); }

inline uint32_t
lind_pick_width (const JobDev &d, bool fused = false)
{
    uint32_t best_w = 0;
    double best = 0.0;
    const uint32_t kmax = (fused && d.halv_h >= 1) ? LIND_KMAX_NARROW_HOST : LIND_KMAX_HOST;
    for (uint32_t k = 1; k <= kmax; k++)
    {
        /* the widest tile whose span fits k rounds (more than LIND_KMAX_HOST rounds: the one-column flavour only) */
        uint32_t lo = 1, hi = k > LIND_KMAX_HOST ? 64 : 128;
        if (lind_span (d, lo) > 64u * k)
            continue;
        while (lo < hi)
        {
            const uint32_t mid = (lo + hi + 1) / 2;
            if (lind_span (d, mid) <= 64u * k)
                lo = mid;
            else
                hi = mid - 1;
        }
        /* slots used in the conversion rounds x lanes used by the filters, both of which scale the tile's cost; a
         * magnifying axis converts few pixels per tile: there the lanes the filters keep busy decide alone */
        const double eff = ((double) lind_span (d, lo) / (64.0 * k)) * ((double) lo / (lo > 64 ? 128.0 : 64.0));
        /* the one-column flavour's candidates (fused, two halvings): columns per unit of work -- a conversion round costs a
         * lane about a third of what its column's filters and packer do */
        const double per_work = (double) lo / (20.0 * k + 60.0 * (lo > 64 ? 2.0 : 1.0));
        const double score = d.step_x < ((uint64_t) 1 << 32) ? (double) lo : (kmax > LIND_KMAX_HOST ? per_work : eff);
        if (score > best + 1e-9)
        {
            best = score;
            best_w = lo;
        }
    }
    if (const char *e = tune_env ("SMOL_LIND_W"))
        best_w = (uint32_t) atoi (e);
    if (best_w < 16 || best_w > 128 || lind_span (d, best_w) > 64u * ((fused && d.halv_h >= 1 && best_w <= 64) ? LIND_KMAX_NARROW_HOST : LIND_KMAX_HOST))
        return 0;
    return best_w;
}

/* KC_DOWN = k_lind, both axes in one kernel: every 128bpp-class job without a box axis that k_up (magnification on
 * both axes) has not taken -- bilinear minification with 0-2 halvings on both axes (the case the kernel was built for),
 * and the jobs whose axes differ: one minifying and one magnifying, or one of them COPY / ONE. Round 4 first sent
 * those through an intermediate image (k_linh + k_linv, below); the kernel's vertical schedule never cared whether
 * consecutive samples share their source rows, and without the image the mixed rows move 40 % of the bytes. */
inline bool
down_setup (JobDev &d)
{
    if (!d.wide || (!d.linear && !d.p16) || tune_env ("SMOL_NO_DOWN"))
        return false;                           /* (compressed 128bpp never meets a bilinear axis) */
    if (d.p16 && d.bpp_in != 4)
        return false;                           /* (premul16 storage only follows unassociated alpha, 4 bytes in) */
    if (d.filter_h == F_BOX || d.filter_v == F_BOX)
        return false;
    if ((d.filter_h != F_ONE && d.w_in < 2) || (d.filter_v == F_BILINEAR && d.h_in < 2))
        return false;
    if (d.filter_h == F_ONE && d.bpp_in == 3)
        return false;                           /* (a row of three bytes: no dword to load a 24-bit pixel with) */
    const bool mixed = d.filter_h != F_BILINEAR || d.filter_v != F_BILINEAR
                    || d.step_x < ((uint64_t) 1 << 32) || d.step_y < ((uint64_t) 1 << 32);
    if (mixed && tune_env ("SMOL_NO_LIND_MIXED"))
        return false;

    JobDev t = d;
    /* COPY and ONE axes have no positions of their own: a step of one pixel (COPY), of nothing (ONE) */
    if (t.filter_h == F_COPY) { t.frac0_x = 0; t.step_x = (uint64_t) 1 << 32; t.halv_h = 0; }
    if (t.filter_h == F_ONE) { t.frac0_x = 0; t.step_x = 0; t.halv_h = 0; }
    if (t.filter_v == F_COPY || t.filter_v == F_ONE) { t.frac0_y = 0; t.step_y = (uint64_t) 1 << 32; t.halv_v = 0; }

    const uint32_t w = lind_pick_width (t, true);
    if (!w)
        return false;

    /* <= 128 vertical samples per tile. A magnifying vertical axis shares its source rows between output rows: tiles
     * tall enough to reach ~16 of them, so that the two halo rows a tile converts again stay a small share */
    /* (behind two vertical halvings an output row already reads eight source rows: tiles of two -- gamma-correct 4000 x 3000
     * RGB8 -> 571 x 428 102 -> 96 us, 4096^2 -> 683^2 98 -> 88; behind one halving 2 / 3 / 4 / 6 rows within 3 %, without
     * halvings eight stay ahead of 2-6 by 2-25 %: profiles/r05_flat3_ab.txt, last table) */
    uint32_t th = t.halv_v == 2 ? 2u : t.halv_v ? 4u : 8u;
    if (t.filter_v == F_BILINEAR && t.step_y < ((uint64_t) 1 << 32))
    {
        const uint64_t want = ((uint64_t) 14 << 32) / (t.step_y ? t.step_y : 1u);
        th = want > 64u ? 64u : want < 8u ? 8u : (uint32_t) want;
    }
    if (const char *e = tune_env ("SMOL_DOWN_ROWS"))
        th = (uint32_t) atoi (e);
    if (th > (128u >> t.halv_v))
        th = 128u >> t.halv_v;
    if (th < 1)
        th = 1;
    t.group_rows = (uint16_t) w;
    t.span_px = (uint32_t) lind_span (t, w);
    t.tile_rows = (uint16_t) th;
    t.tile_cpl = 2;
    t.up_wide = 1;
    t.down_mode = 3;
    d = t;
    return true;
}

/* tiles of the job (a workgroup takes lind_waves () of them) */
inline uint32_t
down_blocks (const JobDev &d)
{
    return ((d.w_out + d.group_rows - 1u) / d.group_rows) * ((d.n_rows + d.tile_rows - 1u) / d.tile_rows);
}

inline uint32_t lind_waves (const JobDev &) { return 4u; }     /* waves (= tiles) per workgroup, around one copy of the tables */

inline uint32_t
down_lds_bytes (const JobDev &d)
{
    return DOWN_LUT_BYTES + lind_waves (d) * (64u * (lind_narrow (d) ? LIND_KMAX_NARROW_HOST : LIND_KMAX_HOST) * (d.p16 ? 16u : 8u)) + 16u;        /* the tables + a row buffer per wave */
}

/* host twin of the device box_span (smol_kernels.hip.h; closed form of smolscale.c's box precalc) */
inline void
box_span_host (uint64_t step, uint32_t dim_in, uint32_t k, uint32_t &o0, uint32_t &o1)
{
    const uint64_t b = (uint64_t) k * step;
    const uint64_t e = b + step;
    const uint32_t last = (uint32_t) (e >> 16);

    o0 = (uint32_t) (b >> 16);
    o1 = last > dim_in - 1 ? dim_in - 1 : last;
}

/* k_boxh + k_boxv: box on both axes in two passes through an intermediate image.
 * Every box/box job of the 128bpp class, and 64bpp-class ones from a vertical ratio
 * of 12 up (below that the single-pass k_tile has enough tiles and wins). */
/* k_boxh: source rows a wave takes, one after the other through ONE LDS row buffer (no prefetch inside the
 * wave: the 32 resident waves per CU -- twice what two buffers per wave allow -- cover each other's
 * latency). 64bpp class: 1 row (4000x3000 -> 200x150: 117-123 us; 4 rows with a second buffer 125-130, 8 rows
 * 129-137, 32 rows 168); 128bpp class: 4 rows, so that a workgroup's 7 KiB table copy serves 16 rows
 * (profiles/r02q_boxh_rows.txt). SMOL_BOXH_ROWS / SMOL_BOXH_ROWS_WIDE: build-time overrides for sweeps. */
#ifndef SMOL_BOXH_ROWS
#define SMOL_BOXH_ROWS 1
#endif
#ifndef SMOL_BOXH_ROWS_WIDE
#define SMOL_BOXH_ROWS_WIDE 4
#endif
inline uint32_t boxh_rows (const JobDev &d) { return d.wide ? SMOL_BOXH_ROWS_WIDE : SMOL_BOXH_ROWS; }

inline void boxh_geometry (JobDev &d, uint32_t lpc, uint64_t span_px, uint32_t chunks);

/* k_boxh's lanes per output column: 64 columns per wave, or 8 columns with 8 lanes each when a column's span is very
 * wide (then 64 columns' worth of a row would not fit the LDS either), or -- spans of hundreds of pixels, which only the
 * 128bpp class has (boxes wider than 255 pixels) -- the whole wave on ONE column: eight columns' spans in a wave's row
 * buffer are 32 KB at a ratio of 1024, one wave per SIMD and nobody to cover its loads (65535 x 512 -> 64 x 8: 295 us) */
inline uint64_t
boxh_lpc8_min ()
{
    if (const char *e = tune_env ("SMOL_BOX2_LPC8_MIN"))
        return (uint64_t) atoi (e);
    return 48;
}

inline uint32_t
boxh_lpc (const JobDev &d)
{
    uint64_t lpc64_min = 400;
    if (const char *e = tune_env ("SMOL_BOX2_LPC64_MIN"))
        lpc64_min = (uint64_t) atoi (e);
    if (d.wide && d.step_x > (lpc64_min << 16))
        return 64u;
    /* from a ratio of 96, sixteen: four columns' spans per row buffer instead of eight (8192 -> 64 gamma-correct 235 -> 208 us,
     * 8192 -> 30 103 -> 101; 32 lanes per column lose to their longer reduction: profiles/r04_boxh_tail_ab.txt) */
    uint64_t lpc16_min = 96;
    if (const char *e = tune_env ("SMOL_BOX2_LPC16_MIN"))
        lpc16_min = (uint64_t) atoi (e);
    if (d.wide && d.step_x > (lpc16_min << 16))
        return 16u;
    return (d.step_x > (boxh_lpc8_min () << 16)) ? 8u : 1u;
}

/* k_boxh without a row buffer (smol_kernel_box2.hip.h: boxh_direct_body): the 128bpp class's whole-wave-per-column flavour */
inline bool
boxh_direct_ok (const JobDev &d, uint32_t lpc)
{
    return d.wide && lpc == 64u && !tune_env ("SMOL_NO_BOXH_DIRECT");
}

inline bool
box2_setup (JobDev &d)
{
    if (d.filter_h != F_BOX || d.filter_v != F_BOX || tune_env ("SMOL_NO_BOX2") || d.n_rows == 0)
        return false;
    /* 64bpp class: the single-pass k_tile or the two passes? k_tile has one wave per 64 columns x a few rows -- a few
     * hundred waves per image at these ratios, each a chain of row groups -- and takes 45-75 us for ONE 4096 x 3072 image
     * whatever the ratio, where the two passes (one wave per 64 columns and source row) take 17-20; with a launch's worth
     * of images in flight it is the faster one up to a ratio of 24 (32-bit sources; 8 lanes per column beyond that do not
     * pay) or 19 (24-bit sources): 8 frames at ratio 9: 96 against 126 us, 10 frames at 16-24: 82-97 against 99-102,
     * RGB8 at 12-17: 72-80 against 80-98 (profiles/r04_box_threshold.txt). So: by the source bytes of the launch. */
    if (!d.wide)
    {
        const uint64_t ratio16 = d.step_y;          /* 16.16 */
        bool tile;
        if (d.box_force)
            tile = d.box_force == 1;
        else if (const char *e = tune_env ("SMOL_BOX2_MIN"))
            tile = ratio16 < ((uint64_t) atoi (e) << 16);
        else
        {
            const uint64_t own_mb = ((uint64_t) d.w_in * d.bpp_in * (((uint64_t) d.n_rows * ratio16) >> 16)) >> 20;
            const uint64_t mb = own_mb > d.launch_src_mb ? own_mb : d.launch_src_mb;
            const uint64_t r12 = (uint64_t) 12 << 16, r16 = (uint64_t) 16 << 16, r19 = (uint64_t) 19 << 16, r24 = (uint64_t) 24 << 16;
            /* (32-bit sources between 12 and 16: k_tile's box loop reads 114-136 us there against the two passes' 108-116) */
            const bool ratio_ok = d.bpp_in == 3 ? ratio16 < r19 : (ratio16 < r12 || (ratio16 >= r16 && ratio16 <= r24));
            tile = ratio_ok && mb >= (ratio16 < r12 ? 150u : 300u);
        }
        if (tile)
            return false;
    }

    uint32_t y0, y1, t;
    box_span_host (d.step_y, d.h_in, d.first_row, y0, t);
    box_span_host (d.step_y, d.h_in, d.first_row + d.n_rows - 1, t, y1);

    /* pass 1: 64 columns per wave, or 8 columns with 8 lanes each when a column's span is very
     * wide (then 64 columns' worth of a row would not fit the LDS either) */
    const uint64_t lpc8_min = boxh_lpc8_min ();
    const uint32_t lpc = boxh_lpc (d);

    /* source bytes a tile can span in one row, as 16-byte chunks (+ misalignment and reader slack) */
    uint64_t span_px = (((64u / lpc) * d.step_x) >> 16) + 2;
    if (span_px > d.w_in)
        span_px = d.w_in;
    const uint32_t bytes = (uint32_t) span_px * d.bpp_in + 15 + 8;
    uint32_t chunks = (bytes + 15) / 16;
    if (chunks * 16u * 4u + DOWN_LUT_BYTES_FWD + 64u > 160u * 1024u)
    {
        /* four waves' row buffers must fit the LDS (nothing of d touched yet). A span beyond that -- more than ~9 600 pixels
         * per column, the whole wave on one column -- is read straight from memory (boxh_direct_body, round 6): no chunks */
        if (!boxh_direct_ok (d, lpc))
            return false;
        chunks = 0;
    }

    boxh_geometry (d, lpc, span_px, chunks);
    d.box2_vlpc = (d.step_y > (lpc8_min << 16)) ? 8 : 1;
    d.tmp_row0 = y0;
    d.tmp_row1 = y1;
    d.box2 = 1;
    return true;
}

/* what k_boxh needs to know about its tiles (box2_setup, twopass_setup) */
inline void
boxh_geometry (JobDev &d, uint32_t lpc, uint64_t span_px, uint32_t chunks)
{
    d.tile_lpc = (uint8_t) lpc;
    {
        /* pass-1 workgroup shape: 4 column tiles side by side (whole rows, contiguous: 4000x3000 -> 200x150
         * 122 us against 128) unless that leaves wave slots empty -- 7 column tiles (4096 -> 448): one below the
         * other 261 us, side by side 300 (profiles/r02q_boxh_rows.txt) */
        const uint32_t tiles_x = (d.w_out + 64u / lpc - 1u) / (64u / lpc);
        d.box2_ygroup = (tiles_x % 4u) ? 1 : 0;
        if (tune_env ("SMOL_BOXH_XGROUP")) d.box2_ygroup = 0;
        if (tune_env ("SMOL_BOXH_YGROUP")) d.box2_ygroup = 1;
    }
    d.span_px = (uint32_t) span_px;
    d.span_chunks = chunks;
    d.span_chunks_inv = chunks ? (uint32_t) (((uint64_t) 1 << 32) / chunks) + 1 : 0u;
    /* most interior pixels a lane walks: a column's span o0 .. o1 has o1 - o0 <= floor (step) + 1, its interior one less */
    d.box_nmax = ((uint32_t) (d.step_x >> 16) + lpc - 1) / lpc;
}

inline bool
box2_is (const JobDev &d)
{
    return d.box2 != 0;
}

/* box2 == 3 (twopass_setup): a bilinear (COPY, ONE) horizontal axis under a TALL vertical box. The intermediate image of
 * the two-pass road would be larger than the source (300 x 20000 -> 200 x 40: 32 MB written and read back for 24 MB of
 * source; 64 x 60000 -> 100 x 4 three times the source) and its second pass a handful of waves walking thousands of rows.
 * Instead pass 1 (k_linh's staging + horizontal filter) keeps the vertical box sums in registers -- the order of the
 * reference's passes stands, every row is filtered horizontally and rounded before it is summed (smolscale-generic.c:
 * 2163-2260) -- with each output row's span of source rows cut into box_nmax SPLITS, one tile (wave) per (column tile,
 * output row, split): 32-bit sums are associative. What reaches memory is one 16-byte partial sum per output pixel and
 * split; pass 2 (k_boxv) adds the splits, normalises and packs. */
SMOL_HD inline bool twopass_hbox (const JobDev &d) { return d.box2 == 3; }
SMOL_HD inline uint32_t hbox_splits (const JobDev &d) { return d.box_nmax; }

/* bytes per pixel of the intermediate image */
inline uint32_t
box2_tmp_px (const JobDev &d)
{
    if (twopass_hbox (d))
        return 16u;
    return (d.box2 == 2 && d.p16 && d.linear && d.filter_h != F_BOX && d.filter_v != F_BOX) ? 16u : 8u;
}

inline size_t
box2_tmp_bytes (const JobDev &d)
{
    if (twopass_hbox (d))
        return (size_t) d.n_rows * hbox_splits (d) * d.w_out * 16u;
    return (size_t) (d.tmp_row1 - d.tmp_row0 + 1) * d.w_out * box2_tmp_px (d);
}

/* where this strip's part of the intermediate image starts in a buffer that holds the whole image's (the plan's ring
 * slots: strips of one image share a slot and must not overlap) */
inline size_t
box2_tmp_offset (const JobDev &d)
{
    if (twopass_hbox (d))
    {
        /* the splits follow the call's row count (twopass_setup), so strips of one image may differ in theirs: rows are
         * placed by the most splits ANY row range of this job takes -- min (64, span / 32) -- and disjoint row ranges stay
         * disjoint in the buffer whatever each of them picked (h_out * span / 32 * 16 bytes <= h_in / 2 per column: inside
         * the plan's h_in * w_out * 8) */
        uint32_t m = (uint32_t) (d.step_y >> 16) / 32u;
        if (m > 64u) m = 64u;
        if (m < hbox_splits (d)) m = hbox_splits (d);       /* (tuning builds: SMOL_HBOX_SPLITS) */
        return (size_t) d.first_row * m * d.w_out * 16u;
    }
    return (size_t) d.tmp_row0 * d.w_out * box2_tmp_px (d);
}

/* tiles (= waves) of each pass; a workgroup runs 4 of them */
inline bool twopass_lin_h (const JobDev &d) { return (d.box2 == 2 || d.box2 == 3) && d.filter_h != F_BOX; }
inline bool twopass_lin_v (const JobDev &d) { return d.box2 == 2 && d.filter_v != F_BOX; }
/* k_linh: row tiles of a job (rows of the intermediate image in groups of tile_rows; fused box: output rows x splits) */
inline uint32_t
linh_row_tiles (const JobDev &d)
{
    return twopass_hbox (d) ? d.n_rows * hbox_splits (d) : (d.tmp_row1 - d.tmp_row0 + d.tile_rows) / d.tile_rows;
}

inline uint32_t
box2_h_tiles (const JobDev &d)
{
    if (twopass_lin_h (d))
        return ((d.w_out + d.group_rows - 1u) / d.group_rows) * linh_row_tiles (d);
    const uint32_t cols = 64u / d.tile_lpc;
    return ((d.w_out + cols - 1u) / cols) * ((d.tmp_row1 - d.tmp_row0 + boxh_rows (d)) / boxh_rows (d));
}

/* pass-1 grid: the 4 waves of a workgroup sit side by side on the same rows (x = groups of 4 column tiles,
 * y = row groups) when the column tiles come in fours, else one below the other on the same columns
 * (JobDev::box2_ygroup) -- packed (x | y << 16) so that a batch can carry the per-axis maxima in one word */
inline uint32_t
box2_h_grid (const JobDev &d)
{
    if (twopass_lin_h (d))      /* k_linh: x = column tiles, y = groups of 4 row tiles (one per wave) */
        return ((d.w_out + d.group_rows - 1u) / d.group_rows) | (((linh_row_tiles (d) + 3u) / 4u) << 16);
    const uint32_t cols = 64u / d.tile_lpc;
    uint32_t gx = (d.w_out + cols - 1u) / cols;
    uint32_t gy = (d.tmp_row1 - d.tmp_row0 + boxh_rows (d)) / boxh_rows (d);
    if (d.box2_ygroup)
        gy = (gy + 3u) / 4u;
    else
        gx = (gx + 3u) / 4u;
    return gx | (gy << 16);
}

inline uint32_t
box2_grid_max (uint32_t a, uint32_t b)
{
    const uint32_t x = (a & 0xffffu) > (b & 0xffffu) ? (a & 0xffffu) : (b & 0xffffu);
    const uint32_t y = (a >> 16) > (b >> 16) ? (a >> 16) : (b >> 16);
    return x | (y << 16);
}

inline uint32_t
box2_v_tiles (const JobDev &d)
{
    if (twopass_lin_v (d))      /* k_linv: 128 columns x tile_rows output rows per wave */
        return ((d.w_out + 127u) / 128u) * ((d.n_rows + d.tile_rows - 1u) / d.tile_rows);
    const uint32_t cols = 64u / d.box2_vlpc;
    return ((d.w_out + cols - 1u) / cols) * d.n_rows;
}

inline uint32_t
box2_h_lds_bytes (const JobDev &d)
{
    if (twopass_lin_h (d))
        return DOWN_LUT_BYTES + 4u * (64u * LIND_KMAX_HOST * (d.p16 ? 16u : 8u)) + 16u;
    return (d.wide ? DOWN_LUT_BYTES_FWD : 0u) + 4u * d.span_chunks * 16u + 32u;
}

/* ---- the generic two-pass road of the 128bpp class (box2 == 2) ------------------------------------------------
 * 128bpp-class jobs whose axes differ -- one minified and one magnified, a COPY axis, bilinear on one axis and box
 * on the other -- used to end in the correctness backstops (k_wtile, k_general<true>: 5-14 % of the roofline,
 * profiles/r04_backstop_rows.txt). No proportional resize lands there, independently scaled axes do. They now take
 * the intermediate image of the two-pass box filter with one kernel per axis:
 *   pass 1   horizontal: k_boxh (BOX) or k_linh = the staging + horizontal half of k_lind (BILINEAR up or down, COPY)
 *   pass 2   vertical:   k_boxv (BOX) or k_linv = the vertical + packing half of k_lind (BILINEAR up or down, COPY)
 * A horizontal COPY axis is a bilinear axis at step 1 (every weight is 256 or, at the last sample, 0: the tap returns
 * the pixel itself); a vertical one hands row R of the intermediate image to output row R. Intermediate pixels: 8 bytes
 * (four 16-bit channels) everywhere but for premul16 + sRGB between two k_lin* passes (19-bit channels: 16 bytes). Gamma-correct and premul16 jobs alike
 * (the 128bpp class); compressed 128bpp (boxes wider than 255 pixels without sRGB) stays with its own kernels. */
inline void
twopass_v_rows (const JobDev &d, uint32_t &y0, uint32_t &y1)
{
    if (d.filter_v == F_BOX)
    {
        uint32_t t;
        box_span_host (d.step_y, d.h_in, d.first_row, y0, t);
        box_span_host (d.step_y, d.h_in, d.first_row + d.n_rows - 1, t, y1);
        return;
    }
    if (d.filter_v == F_COPY)
    {
        y0 = d.first_row;
        y1 = d.first_row + d.n_rows - 1u;
        return;
    }
    if (d.filter_v == F_ONE)
    {
        y0 = y1 = 0;
        return;
    }
    /* bilinear: sample j reads rows o (j) and o (j) + 1 */
    auto ofs = [&] (uint64_t j) {
        const uint64_t o = (d.frac0_y + j * d.step_y) >> 32;
        return (uint32_t) (o >= d.h_in - 1u ? d.h_in - 2u : o);
    };
    y0 = ofs ((uint64_t) d.first_row << d.halv_v);
    y1 = ofs ((((uint64_t) d.first_row + d.n_rows) << d.halv_v) - 1u) + 1u;
}

inline bool
twopass_setup (JobDev &d)
{
    if (!d.wide || d.n_rows == 0 || tune_env ("SMOL_NO_TWOPASS"))
        return false;
    /* (neither sRGB nor premul16: the class's compressed storage, which a box of 256+ pixels on one axis brings -- 8-bit
     * values that the 11-bit passes carry as they are) */
    if (!d.linear && !d.p16 && tune_env ("SMOL_NO_TWOPASS_FLAT"))
        return false;
    if (d.p16 && d.bpp_in != 4)
        return false;                           /* (premul16 storage only follows unassociated alpha, 4 bytes in) */
    /* premul16 + sRGB leaves 19-bit channels behind a bilinear horizontal pass: more than the intermediate image's 16-bit
     * pixels hold in front of k_boxv. With the vertical box sums kept in pass 1's registers (box2 == 3, below) nothing is
     * narrowed: that road takes these jobs at every box ratio (round 5; k_general<true> before, 3.4 % of random jobs) */
    const bool p16l_bilbox = d.p16 && d.linear && d.filter_h != F_BOX && d.filter_v == F_BOX;
    if (p16l_bilbox && (tune_env ("SMOL_NO_HBOX") || (uint64_t) d.n_rows >= 4u * 65535u))
        return false;
    if (d.filter_h == F_BOX && d.filter_v == F_BOX)
        return false;                           /* box2_setup's */
    if ((d.filter_h == F_BILINEAR && d.w_in < 2) || (d.filter_v == F_BILINEAR && d.h_in < 2))
        return false;
    if ((d.filter_h == F_COPY && d.w_in < 2))
        return false;
    if (d.filter_h == F_ONE && d.bpp_in == 3)
        return false;                           /* (a row of three bytes: no dword to load a 24-bit pixel with) */

    JobDev t = d;
    /* COPY and ONE axes have no positions of their own: a step of one pixel (COPY), of nothing (ONE) */
    if (t.filter_h == F_COPY) { t.frac0_x = 0; t.step_x = (uint64_t) 1 << 32; t.halv_h = 0; }
    if (t.filter_h == F_ONE) { t.frac0_x = 0; t.step_x = 0; t.halv_h = 0; }
    if (t.filter_v == F_COPY || t.filter_v == F_ONE) { t.frac0_y = 0; t.step_y = (uint64_t) 1 << 32; t.halv_v = 0; }

    if (t.filter_h == F_BOX)
    {
        /* k_boxh's geometry, as box2_setup makes it */
        const uint32_t lpc = boxh_lpc (t);
        uint64_t span_px = (((64u / lpc) * t.step_x) >> 16) + 2;
        if (span_px > t.w_in)
            span_px = t.w_in;
        uint32_t chunks = ((uint32_t) span_px * t.bpp_in + 15 + 8 + 15) / 16;
        if (chunks * 16u * 4u + DOWN_LUT_BYTES_FWD + 64u > 160u * 1024u)
        {
            if (!boxh_direct_ok (t, lpc))
                return false;
            chunks = 0;                         /* (straight from memory: boxh_direct_body) */
        }
        boxh_geometry (t, lpc, span_px, chunks);
    }
    else
    {
        /* k_linh: k_lind's tile width */
        const uint32_t best_w = lind_pick_width (t);
        if (!best_w)
            return false;
        t.group_rows = (uint16_t) best_w;
        t.span_px = (uint32_t) lind_span (t, best_w);
    }
    t.box2_vlpc = (t.filter_v == F_BOX && t.step_y > ((uint64_t) 48 << 16)) ? 8 : 1;
    t.tile_rows = 8;                            /* rows per tile of both k_lin* passes: at most 8 << 2 vertical samples */
    t.tile_cpl = 2;
    twopass_v_rows (t, t.tmp_row0, t.tmp_row1);
    t.box2 = 2;
    /* a vertical box behind a bilinear (COPY, ONE) horizontal axis: the vertical sums stay in pass 1's registers (box2 == 3,
     * above).
     * At most 64 splits and the grid's y limit (65535 workgroups of 4 row tiles). */
    /* (at every box ratio: 4096^2 -> 2731 x 300, V ratio 13.7: gamma-correct 95 -> 63 us, premul16 98 -> 47, premul16 + gamma
     * 83 (k_general) -> 57; SMOL_HBOX_MIN in tuning builds: the old two-pass road below that ratio) */
    uint32_t hbox_min = 0;
    if (const char *e = tune_env ("SMOL_HBOX_MIN"))
        hbox_min = (uint32_t) atoi (e);
    if (t.filter_h != F_BOX && t.filter_v == F_BOX && ((t.step_y >> 16) >= hbox_min || p16l_bilbox) && !tune_env ("SMOL_NO_HBOX")
        && (uint64_t) t.n_rows < 4u * 65535u)
    {
        /* enough splits for ~2048 tiles per image (the chip holds ~7000 waves), each at least 32 source rows long */
        const uint32_t tiles = ((t.w_out + t.group_rows - 1u) / t.group_rows) * t.n_rows;
        const uint32_t span = (uint32_t) (t.step_y >> 16);
        uint32_t sp = (2048u + tiles - 1u) / tiles;
        if (sp > span / 32u) sp = span / 32u;       /* (short spans: one tile per output row) */
        if (const char *e = tune_env ("SMOL_HBOX_SPLITS"))
            sp = (uint32_t) atoi (e);
        if (sp > 64u) sp = 64u;                 /* (64 x 60000 -> 100 x 4: 64 splits 161 us, 128 157, 256 183) */
        if (t.hbox_force)
            sp = t.hbox_force;                  /* (the strip's choice: strip_classify) */
        while (sp > 1u && (uint64_t) t.n_rows * sp > 4u * 65535u)
            sp--;
        if (sp < 1u) sp = 1u;
        t.box_nmax = sp;
        t.box2_vlpc = 1;
        t.box2 = 3;
    }
    d = t;
    return true;
}

/* k_hspan (smol_kernel_hspan.hip.h): vertical COPY, horizontal bilinear at (nearly) the power-of-two
 * ratio its halvings imply -- at most two places per row where the sampling grid slips by a pixel
 * (BASELINE config 4: 65535 -> 8192, BILINEAR_2H, one slip). 64bpp class, any pixel types; 24-bit
 * input needs at least one halving (a lane's window must be a multiple of 16 bytes). */
/* rows a workgroup walks. ONE: workgroups then sweep the image in raster order, which is what HBM
 * wants -- config 4 takes 150 us with 1 row per workgroup, 175 with 2, 188 with 4, 197 with 8, 292
 * with 64 (tools/microbench/hspan_ab.hip, profiles/r02c_hspan_ab.txt); the per-row set-up is 3 % */
constexpr uint32_t HSPAN_BAND_ROWS = 1;

inline bool
hspan_setup (JobDev &d)
{
    if (d.wide || d.filter_v != F_COPY || d.filter_h != F_BILINEAR || tune_env ("SMOL_NO_HSPAN"))
        return false;
    if (d.bpp_in == 3 && d.halv_h == 0)
        return false;
    const uint64_t full = (uint64_t) d.w_out << (d.halv_h + 1);
    const uint64_t diff = full > d.w_in ? full - d.w_in : d.w_in - full;
    if (diff > 2 || d.w_out < 256 || d.w_in < 4)
        return false;
    d.tile_rows = (uint16_t) HSPAN_BAND_ROWS;
    if (const char *e = tune_env ("SMOL_HSPAN_ROWS"))
        if (atoi (e) >= 1 && atoi (e) <= 4096)
            d.tile_rows = (uint16_t) atoi (e);
    return true;
}

inline uint32_t
hspan_blocks (const JobDev &d)
{
    return ((d.w_out + 1023u) / 1024u) * ((d.n_rows + d.tile_rows - 1u) / d.tile_rows);
}

inline uint32_t
hspan_lds_bytes (const JobDev &d)
{
    const uint32_t wb = 4u * (2u << d.halv_h) * d.bpp_in;           /* bytes of a lane's window */
    const uint32_t nl = (64u * wb + 1023u) / 1024u + 1u;
    return 4u * (nl * 1024u + 32u);
}

inline KClass
classify (JobDev &d)
{
    if (pow2_ok (d))
        return KC_POW2;
    if (pow2x_setup (d))
        return KC_POW2X;
    if (up_setup (d))
        return KC_UP;
    if (down_setup (d))
        return KC_DOWN;
    if (box2_setup (d))
        return KC_BOX2;
    if (twopass_setup (d))
        return KC_BOX2;
    if (hspan_setup (d))
        return KC_HSPAN;
    if (tile_ok (d))
        return KC_TILE;
    if (wtile_ok (d))
        return KC_WTILE;
    /* what is left is the 128bpp class with a box axis (or a minifying one). A tall
     * vertical box over few output pixels: one wave per pixel instead of one lane */
    if (d.filter_v == F_BOX && (d.step_y >> 16) >= 32 && (uint64_t) d.w_out * d.n_rows <= 262144)
        return d.wide ? KC_WAVEPX_WIDE : KC_WAVEPX;
    return d.wide ? KC_GENERAL_WIDE : KC_GENERAL;
}

/* A strip of rows that the host road (smol_rt_lanes.hip) cuts into chunks of ~8 MB: the class is decided ONCE, on the
 * whole strip. 64bpp-class box x box picks the single pass or the two passes by the bytes a launch reads (box2_setup);
 * a chunk classified on its own rows would take the two passes where the strip's probe said k_tile and had sized no
 * intermediate image (8000 x 6000 -> 800 x 600). The verdict travels to the chunks as
 * JobDev::box_force; tmp_bytes is what the strip's intermediate image needs (0: none), an upper bound of any chunk's.
 * Likewise the splits of the bilinear x tall-box road (box2 == 3): twopass_setup takes more splits the fewer rows a call
 * has, so a chunk on its own would need more partial sums per row than the strip's probe sized (96 x 32768
 * -> 400 x 512 gamma-correct from pageable memory, 1 split for the strip, 2 for its 320-row chunks). The strip's count
 * travels as JobDev::hbox_force; rows x splits x w_out x 16 then bounds every chunk. */
struct StripClass
{
    KClass k;
    size_t tmp_bytes;
    uint8_t box_force;
    uint8_t hbox_force;
};

/* hand a strip's verdicts to one of its chunks */
inline void
strip_force (JobDev &jd, const StripClass &sc)
{
    jd.box_force = sc.box_force;
    jd.hbox_force = sc.hbox_force;
}

inline StripClass
strip_classify (const SmolHipJob &job, uint32_t src_pitch, uint32_t dst_pitch, uint32_t first_row, uint32_t n_rows)
{
    JobDev probe;
    StripClass sc = { KC_GENERAL, 0, 0, 0 };

    fill_jobdev (probe, job, reinterpret_cast<const void *> (uintptr_t (16)), src_pitch,
                 reinterpret_cast<void *> (uintptr_t (16)), dst_pitch, first_row, n_rows);
    sc.k = classify (probe);
    if (sc.k == KC_BOX2)
        sc.tmp_bytes = box2_tmp_bytes (probe);
    if (!probe.wide && probe.filter_h == F_BOX && probe.filter_v == F_BOX)
        sc.box_force = sc.k == KC_BOX2 ? 2 : 1;
    if (sc.k == KC_BOX2 && twopass_hbox (probe))
        sc.hbox_force = (uint8_t) hbox_splits (probe);      /* (1 .. 64) */
    return sc;
}

inline const char *
kclass_name (KClass k, const JobDev &d)
{
    switch (k)
    {
        case KC_POW2: return "k_pow2_batch";
        case KC_POW2X: return "k_pow2x_batch";
        case KC_UP: return d.up_wide >= 2 ? "k_upg_batch" : "k_up_batch";
        case KC_DOWN: return "k_lind_batch";
        case KC_BOX2: return d.box2 == 3 ? "k_linh_batch (vertical box sums in registers) + k_boxv_batch"
                             : d.box2 != 2 ? "k_boxh_batch + k_boxv_batch"
                             : twopass_lin_h (d) ? (twopass_lin_v (d) ? "k_linh_batch + k_linv_batch" : "k_linh_batch + k_boxv_batch")
                                                 : "k_boxh_batch + k_linv_batch";
        case KC_HSPAN: return "k_hspan_batch";
        case KC_TILE: return "k_tile_batch";
        case KC_WTILE: return "k_wtile_batch";
        case KC_WAVEPX: return "k_wavepx_batch<false>";
        case KC_WAVEPX_WIDE: return "k_wavepx_batch<true>";
        case KC_GENERAL_WIDE: return "k_general_batch<true>";
        default: return "k_general_batch<false>";
    }
}

} /* namespace dispatch */
} /* namespace smolhip */
