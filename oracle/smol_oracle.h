/* smol_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, per-channel integer arithmetic) of the reference's
 * resampling path: planner + unpack + horizontal/vertical filters + pack.
 * It exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg can CHECK the HIP product path; nothing under smolscale_amd/ may include,
 * link or call it.
 *
 * Parity status: PINNED. tests/test_oracle_vs_reference.py compares this oracle
 * byte-for-byte with the real reference compiled into oracle/_ref (generic
 * backend) on a randomized sweep of every pixel-type pair x filter class, and
 * tests/golden/ holds vectors captured from that reference build. */

#ifndef SMOL_ORACLE_H
#define SMOL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Filter ids use the reference's numbering (smolscale-private.h:100-116) so the
 * planner can be compared field by field with ref_probe_plan(). */
enum
{
    ORACLE_FILTER_COPY = 0,
    ORACLE_FILTER_ONE = 1,
    ORACLE_FILTER_BILINEAR_0H = 2,
    ORACLE_FILTER_BILINEAR_1H = 3,
    ORACLE_FILTER_BILINEAR_2H = 4,
    ORACLE_FILTER_BOX = 9
};

typedef struct
{
    uint32_t filter_h, filter_v;          /* reference enum values              */
    uint32_t storage_type;                /* 2 = 64bpp, 3 = 128bpp (ref enum)   */
    uint32_t gamma_type;                  /* 0 = compressed, 1 = linear         */
    uint32_t width_bilin_out, height_bilin_out;
    uint32_t width_halvings, height_halvings;
    uint32_t span_mul_x, span_mul_y;
    /* Not visible in the reference ctx (derived from its repack choice): */
    uint32_t premul16;                    /* unassoc -> unassoc                 */
    uint32_t quirk24;                     /* 24bpp+sRGB out through the `123' packer */
}
OraclePlan;

/* Planner only. Returns 0 on success, -1 on invalid arguments. */
int smol_oracle_plan (uint32_t type_in, uint32_t w_in, uint32_t h_in,
                      uint32_t type_out, uint32_t w_out, uint32_t h_out,
                      uint32_t with_srgb, OraclePlan *plan_out);

/* Precalc arrays in the reference's own layout (uint16 pairs; x: relative
 * offsets / whole-pixel counts, y: absolute). Returns #uint16 written. */
uint32_t smol_oracle_precalc (uint32_t w_in, uint32_t h_in, uint32_t w_out, uint32_t h_out,
                              uint32_t axis, uint16_t *dest, uint32_t max_u16);

/* Scales output rows [first_row, first_row + n_rows) of the job into `dest`
 * (row first_row lands at dest + 0, stride rowstride_out) -- the contract of
 * smol_scale_batch_full() (smolscale.h:79-81). Returns 0 on success. */
int smol_oracle_scale_rows (const void *pixels_in, uint32_t type_in,
                            uint32_t w_in, uint32_t h_in, uint32_t rowstride_in,
                            void *dest, uint32_t type_out,
                            uint32_t w_out, uint32_t h_out, uint32_t rowstride_out,
                            uint32_t with_srgb,
                            uint32_t first_row, uint32_t n_rows);

/* Deterministic synthetic input generator shared by tests and bench
 * (xorshift64). If valid_premul_type >= 0 and that pixel
 * type is premultiplied, colour bytes are reduced modulo (alpha + 1). */
void smol_oracle_fill (void *buf, uint64_t n_bytes, uint64_t seed);
void smol_oracle_make_valid_premul (void *pixels, uint32_t type, uint32_t width,
                                    uint32_t height, uint32_t rowstride);

/* FNV-1a 64 over a strided image (only width*bpp bytes of each row). */
uint64_t smol_oracle_fnv1a (const void *pixels, uint32_t row_bytes, uint32_t height,
                            uint32_t rowstride);

#ifdef __cplusplus
}
#endif

#endif
