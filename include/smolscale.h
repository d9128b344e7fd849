/* smolscale.h -- public C API of the MI355X-native resampler.
 *
 * This header is the drop-in contract: it declares the same six entry points,
 * the same SmolPixelType values (0..9) and the same SmolPostRowFunc callback as
 * the reference's public header (reference smolscale.h:16-83), so a program
 * built against the reference links against libsmolscale_amd.so unchanged.
 * The implementation behind it is new: a host-side planner in C
 * (smolscale_amd/csrc/smol_plan.c) and HIP kernels for gfx950 reached through
 * the C-ABI shim declared in smol_hip.h.
 *
 * Semantics kept from the reference:
 *   - rowstrides are in BYTES and need not be multiples of the pixel size;
 *     pixel pointers need no particular alignment;
 *   - the caller owns pixels_in / pixels_out for the lifetime of a context;
 *   - no function returns an error; unrecoverable internal failures abort()
 *     (reference smolscale.c:779-780, 812-813). A missing GPU / failed HIP call
 *     is such a failure here under the default SMOL_BACKEND=hip: nothing falls back
 *     silently. The library's own CPU backend (smol_cpu.h, a correctness fallback)
 *     runs only when asked for: SMOL_BACKEND=cpu, or SMOL_BACKEND=auto, which says
 *     so on stderr once per process when it takes over.
 *   - results are bit-identical to the reference's generic backend. */

#ifndef SMOLSCALE_AMD_SMOLSCALE_H
#define SMOLSCALE_AMD_SMOLSCALE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Pixel layouts, named by the order of bytes in memory.
 * Values 0..3: alpha-premultiplied; 4..7: unassociated alpha; 8..9: no alpha. */
typedef enum
{
    SMOL_PIXEL_RGBA8_PREMULTIPLIED = 0,
    SMOL_PIXEL_BGRA8_PREMULTIPLIED = 1,
    SMOL_PIXEL_ARGB8_PREMULTIPLIED = 2,
    SMOL_PIXEL_ABGR8_PREMULTIPLIED = 3,

    SMOL_PIXEL_RGBA8_UNASSOCIATED  = 4,
    SMOL_PIXEL_BGRA8_UNASSOCIATED  = 5,
    SMOL_PIXEL_ARGB8_UNASSOCIATED  = 6,
    SMOL_PIXEL_ABGR8_UNASSOCIATED  = 7,

    SMOL_PIXEL_RGB8                = 8,
    SMOL_PIXEL_BGR8                = 9,

    SMOL_PIXEL_MAX                 = 10
}
SmolPixelType;

/* Called once per finished output row (packed pixels, `width' of them), in
 * ascending row order within one batch call; may edit the row in place.
 * With the HIP backend it runs on the host after the strip has been copied
 * back from the device. */
typedef void (SmolPostRowFunc) (uint32_t *row_inout, int width, void *user_data);

typedef struct SmolScaleCtx SmolScaleCtx;

/* One-shot: scale the whole image. Single caller. */
void smol_scale_simple (const void *pixels_in, SmolPixelType pixel_type_in,
                        uint32_t width_in, uint32_t height_in, uint32_t rowstride_in,
                        void *pixels_out, SmolPixelType pixel_type_out,
                        uint32_t width_out, uint32_t height_out, uint32_t rowstride_out,
                        uint8_t with_srgb);

/* Context API: plan once, then produce any ranges of output rows. */
SmolScaleCtx *smol_scale_new (const void *pixels_in, SmolPixelType pixel_type_in,
                              uint32_t width_in, uint32_t height_in, uint32_t rowstride_in,
                              void *pixels_out, SmolPixelType pixel_type_out,
                              uint32_t width_out, uint32_t height_out, uint32_t rowstride_out,
                              uint8_t with_srgb);

SmolScaleCtx *smol_scale_new_full (const void *pixels_in, SmolPixelType pixel_type_in,
                                   uint32_t width_in, uint32_t height_in, uint32_t rowstride_in,
                                   void *pixels_out, SmolPixelType pixel_type_out,
                                   uint32_t width_out, uint32_t height_out, uint32_t rowstride_out,
                                   uint8_t with_srgb,
                                   SmolPostRowFunc post_row_func, void *user_data);

void smol_scale_destroy (SmolScaleCtx *scale_ctx);

/* Produce output rows [first_outrow, first_outrow + n_outrows) into pixels_out.
 * May be called concurrently from several threads on one context as long as the
 * row ranges do not overlap; destroy the context only after all calls return. */
void smol_scale_batch (const SmolScaleCtx *scale_ctx, uint32_t first_outrow, uint32_t n_outrows);

/* Same, but row `first_outrow' is written at outrows_dest + 0 and following rows
 * rowstride_out bytes apart, instead of into pixels_out. */
void smol_scale_batch_full (const SmolScaleCtx *scale_ctx, void *outrows_dest,
                            uint32_t first_outrow, uint32_t n_outrows);

#ifdef __cplusplus
}
#endif

#endif /* SMOLSCALE_AMD_SMOLSCALE_H */
