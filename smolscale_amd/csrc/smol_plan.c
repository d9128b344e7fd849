/* smol_plan.c -- host-side planner for the HIP backend (plain C, no GPU needed).
 *
 * Decides, per axis, which filter class a job uses and derives the fixed-point
 * constants the kernels sample with. The DECISIONS must equal the reference's
 * (pick_filter_params() smolscale.c:427-478, smol_scale_init() :816-870,
 * get_implementations() :725-783) because bit-exact output depends on them;
 * the REPRESENTATION is different: instead of per-sample uint16 tables
 * (smolscale-generic.c:14-135) each axis is described by two 64-bit numbers,
 * so a kernel can compute any sample's source offset and weight in registers. */

#include <string.h>
#include "smol_hip.h"

#define FIX32 ((uint64_t) 1 << 32)

static int
type_is_unassociated (uint32_t t)
{
    return t >= 4 && t <= 7;       /* smolscale.h:16-36, smolscale.c:45-58 */
}

static int
type_is_24bpp (uint32_t t)
{
    return t >= 8;
}

/* One axis. Thresholds from smolscale.c:443-478: ratio > 255 -> box with wide
 * accumulators; ratio > 8 -> box; 1 input sample -> replicate; equal -> copy;
 * otherwise bilinear to out << h followed by h halvings, where h counts the
 * doublings of `out' that stay below `in' (so h <= 2 whenever ratio <= 8). */
static void
plan_axis (uint32_t in, uint32_t out, uint32_t with_srgb,
           uint8_t *filter, uint8_t *halvings, int *wide,
           uint64_t *frac0, uint64_t *step, uint32_t *span_mul)
{
    *halvings = 0;
    *frac0 = 0;
    *step = 0;
    *span_mul = 0;
    *wide = with_srgb ? 1 : 0;

    if (in > out * 255u || in > out * 8u)
    {
        uint64_t whole, part, denom;

        *filter = SMOL_HIP_FILTER_BOX;
        if (in > out * 255u)
            *wide = 1;

        /* 16.16 step and the rounding normaliser of smolscale-generic.c:82-91:
         * span_mul = round (2^24 * 255 / (stride * 255 + floor (f * 255 / 256))) */
        *step = ((uint64_t) in << 16) / out;
        whole = *step >> 16;
        part = (*step >> 8) & 0xff;
        denom = whole * 255 + (part * 255) / 256;
        *span_mul = (uint32_t) ((((uint64_t) 255 << 24) + denom / 2) / denom);
    }
    else if (in == 1)
    {
        *filter = SMOL_HIP_FILTER_ONE;
    }
    else if (in == out)
    {
        *filter = SMOL_HIP_FILTER_COPY;
    }
    else
    {
        uint32_t h = 0, grid = out;

        while ((uint64_t) grid * 2 < in)
        {
            grid *= 2;
            h++;
        }

        *filter = SMOL_HIP_FILTER_BILINEAR;
        *halvings = (uint8_t) h;

        /* 32.32 sampling grid of smolscale-generic.c:24-37: sample centres when
         * shrinking, corner-aligned when enlarging. */
        if (in > grid)
        {
            *step = ((uint64_t) in * FIX32) / grid;
            *frac0 = (*step - FIX32) / 2;
        }
        else
        {
            *step = ((uint64_t) (in - 1) * FIX32) / (grid > 1 ? grid - 1 : 1);
            *frac0 = 0;
        }
    }
}

int
smol_hip_plan_job (uint32_t pixel_type_in, uint32_t width_in, uint32_t height_in,
                   uint32_t pixel_type_out, uint32_t width_out, uint32_t height_out,
                   uint32_t with_srgb, SmolHipJob *job)
{
    int wide_h, wide_v;

    if (!job || pixel_type_in >= 10 || pixel_type_out >= 10
        || width_in == 0 || height_in == 0 || width_out == 0 || height_out == 0
        || width_in > 65535 || height_in > 65535 || width_out > 65535 || height_out > 65535)
        return SMOL_HIP_E_INVALID;

    memset (job, 0, sizeof (*job));
    job->width_in = width_in;
    job->height_in = height_in;
    job->width_out = width_out;
    job->height_out = height_out;
    job->pixel_type_in = (uint8_t) pixel_type_in;
    job->pixel_type_out = (uint8_t) pixel_type_out;

    plan_axis (width_in, width_out, with_srgb, &job->filter_h, &job->halvings_h, &wide_h,
               &job->frac0_x, &job->step_x, &job->span_mul_x);
    plan_axis (height_in, height_out, with_srgb, &job->filter_v, &job->halvings_v, &wide_v,
               &job->frac0_y, &job->step_y, &job->span_mul_y);

    job->wide = (uint8_t) (wide_h | wide_v);                 /* smolscale.c:862 */
    job->linear = with_srgb ? 1 : 0;

    /* unassociated in AND out: 16-bit premultiplied internal form, wide lanes
     * (smolscale.c:751-758) */
    if (type_is_unassociated (pixel_type_in) && type_is_unassociated (pixel_type_out))
    {
        job->premul16 = 1;
        job->wide = 1;
    }

    /* beyond 8191:1 the linear representation would overflow its lanes, so the
     * reference silently stays in compressed sRGB (smolscale.c:760-770); the lane
     * width chosen above is kept. */
    if (width_in > width_out * 8191u || height_in > height_out * 8191u)
        job->linear = 0;

    /* Which 24bpp packer the reference's repack search (smolscale.c:647-719)
     * selects matters only for sRGB-linear jobs: its `123' packer
     * (smolscale-generic.c:922-935) feeds the still-premultiplied value to the
     * sRGB table. The search picks it when the channel order left by the
     * unpacker already equals the output order: inputs whose first colour byte
     * in memory is R (RGBA, ARGB) or the BGR8 type going to BGR8, and the
     * mirrored set going to RGB8. */
    if (type_is_24bpp (pixel_type_out))
    {
        uint32_t t = pixel_type_in;
        int bgr_mid = (t == 0 || t == 2 || t == 4 || t == 6 || t == 9);

        job->quirk24 = (uint8_t) (bgr_mid ? (pixel_type_out == 9) : (pixel_type_out == 8));
    }

    return 0;
}

void
smol_hip_bilinear_sample (uint64_t frac0, uint64_t step, uint32_t dim_in, uint32_t j,
                          uint32_t *ofs, uint32_t *f)
{
    uint64_t pos = frac0 + (uint64_t) j * step;
    uint32_t o = (uint32_t) (pos >> 32);

    /* Past the last pixel pair: pin to it with all the weight on the right
     * neighbour (smolscale-generic.c:43-65). */
    if (o >= dim_in - 1)
    {
        *ofs = dim_in - 2;
        *f = 0;
    }
    else
    {
        *ofs = o;
        *f = 256 - (uint32_t) ((pos >> 24) & 0xff);
    }
}

void
smol_hip_box_span (uint64_t step, uint32_t dim_in, uint32_t k,
                   uint32_t *ofs0, uint32_t *ofs1, uint32_t *f_right, uint32_t *w_left)
{
    uint64_t begin = (uint64_t) k * step;
    uint64_t end = begin + step;
    uint32_t last = (uint32_t) (end >> 16);

    if (last > dim_in - 1)
        last = dim_in - 1;                                   /* smolscale-generic.c:105-110 */

    *ofs0 = (uint32_t) (begin >> 16);
    *ofs1 = last;
    *f_right = (uint32_t) ((end >> 8) & 0xff);
    /* left edge keeps what the previous box did not take: 255 - F (not 256 - F),
     * smolscale-generic.c:1461-1462; the very first pixel counts in full. */
    *w_left = k ? 255 - (uint32_t) ((begin >> 8) & 0xff) : 256;
}

int
smol_hip_input_row_range (const SmolHipJob *job, uint32_t first_row, uint32_t n_rows,
                          uint32_t *in_row_first, uint32_t *in_row_last)
{
    uint32_t a, b, f;

    /* first_row + n_rows <= height_out, written so that it cannot wrap */
    if (!job || n_rows == 0 || first_row >= job->height_out || n_rows > job->height_out - first_row)
        return SMOL_HIP_E_INVALID;

    switch (job->filter_v)
    {
        case SMOL_HIP_FILTER_COPY:
            a = first_row;
            b = first_row + n_rows - 1;
            break;
        case SMOL_HIP_FILTER_ONE:
            a = b = 0;
            break;
        case SMOL_HIP_FILTER_BILINEAR:
            smol_hip_bilinear_sample (job->frac0_y, job->step_y, job->height_in,
                                      first_row << job->halvings_v, &a, &f);
            smol_hip_bilinear_sample (job->frac0_y, job->step_y, job->height_in,
                                      ((first_row + n_rows) << job->halvings_v) - 1, &b, &f);
            b += 1;
            break;
        default:
        {
            uint32_t o0, o1, w;
            smol_hip_box_span (job->step_y, job->height_in, first_row, &a, &o1, &f, &w);
            smol_hip_box_span (job->step_y, job->height_in, first_row + n_rows - 1, &o0, &b, &f, &w);
            break;
        }
    }

    *in_row_first = a;
    *in_row_last = b;
    return 0;
}

const char *
smol_hip_strerror (int err)
{
    switch (err)
    {
        case 0: return "success";
        case SMOL_HIP_E_INVALID: return "invalid argument";
        case SMOL_HIP_E_UNSUPPORTED: return "job not supported by the HIP kernels";
        case SMOL_HIP_E_NO_DEVICE: return "no usable HIP device";
        default: return "HIP runtime error (value is -hipError_t)";
    }
}
