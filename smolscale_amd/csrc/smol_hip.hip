/* smol_hip.hip -- C-ABI shim + host runtime of the HIP backend (gfx950).
 *
 * Exports exactly the functions declared in include/smol_hip.h. Host logic:
 * turning a planner job into a device job descriptor, choosing a kernel class,
 * staging rows between host and device for the host-pointer entry, batching.
 * No CPU compute path exists here: if HIP fails, the error is returned (and the
 * public API in smolscale_api.c aborts, like the reference does on its internal
 * errors, reference smolscale.c:779-780). */

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "smol_hip.h"
#include "smol_jobdev.h"
#include "smol_launch.h"

using smolhip::JobDev;
using namespace smolhip::dispatch;

#define HIP_TRY(expr)                                           \
    do {                                                        \
        hipError_t e_ = (expr);                                 \
        if (e_ != hipSuccess)                                   \
            return -(int) e_;                                   \
    } while (0)

namespace {

/* Launch one kernel over `n_jobs' jobs that all share class k and the per-class
 * grouping keys of smol_hip_batch_create. max_* are maxima over the group. */
hipError_t
launch_class (KClass k, const JobDev *jobs_dev, uint32_t n_jobs, const JobDev &rep,
              uint32_t max_w_out, uint32_t max_rows, hipStream_t s, uint32_t max_lds = 0)
{
    namespace L = smolhip::launch;

    if (k == KC_POW2)
    {
        const uint32_t rx = 2u << rep.halv_h;
        const uint32_t opt = rx >= 4 ? 1 : 4 / rx;
        const uint32_t tcols = max_w_out / opt;
        L::pow2 (jobs_dev, rep, dim3 (((tcols + 63) / 64) * ((max_rows + 15) / 16), n_jobs), s);
    }
    else if (k == KC_POW2X)
    {
        /* max_lds carries the group's largest block count */
        L::pow2x (jobs_dev, rep, dim3 (max_lds ? max_lds : pow2x_blocks (rep), n_jobs), s);
    }
    else if (k == KC_DOWN)
    {
        /* max_lds carries the largest tile count, max_w_out (batches only) the largest LDS size */
        L::down (jobs_dev, rep, dim3 (max_lds ? max_lds : down_blocks (rep), n_jobs),
                 jobs_dev ? max_w_out : down_lds_bytes (rep), s);
    }
    else if (k == KC_UP)
    {
        L::up (jobs_dev, rep, dim3 (max_lds ? max_lds : up_blocks (rep), n_jobs),
               jobs_dev ? max_w_out : up_lds_bytes (rep), s);
    }
    else if (k == KC_BOX2)
    {
        /* rep.tmp / jobs [i].tmp point at the intermediate images; max_w_out / max_rows
         * carry the group's largest pass-1 / pass-2 wave counts, max_lds the largest LDS size */
        L::box2 (jobs_dev, rep, n_jobs, jobs_dev ? max_w_out : box2_h_tiles (rep),
                 jobs_dev ? max_rows : box2_v_tiles (rep), jobs_dev ? max_lds : box2_h_lds_bytes (rep), s);
    }
    else if (k == KC_TILE || k == KC_WTILE)
    {
        /* groups share tile_rows and LDS size */
        const uint32_t tw = 64u * rep.tile_cpl / rep.tile_lpc;
        const uint32_t tiles_x = (max_w_out + tw - 1) / tw;
        const uint32_t tiles_y = (max_rows + rep.tile_rows - 1) / rep.tile_rows;
        /* k_tile deals its tiles out in XCD bands: the count is rounded up to a multiple of 8 */
        const dim3 grid (k == KC_TILE ? ((tiles_x * tiles_y + 7u) & ~7u) : tiles_x * tiles_y, n_jobs);
        const uint32_t lds = max_lds ? max_lds : tile_lds_bytes (rep);

        if (k == KC_WTILE)
            L::wtile (jobs_dev, rep, grid, lds, s);
        else
            L::tile (jobs_dev, rep, grid, lds, s);
    }
    else if (k == KC_WAVEPX || k == KC_WAVEPX_WIDE)
    {
        /* groups hold jobs of one output width */
        L::wavepx (jobs_dev, rep, dim3 (rep.w_out * max_rows, n_jobs), k == KC_WAVEPX_WIDE, s);
    }
    else
    {
        L::general (jobs_dev, rep, dim3 (((max_w_out + 255u) / 256u) * max_rows, n_jobs), k == KC_GENERAL_WIDE, s);
    }
    return hipGetLastError ();
}

/* One job, descriptor by value. The two-pass box filter needs an intermediate image:
 * `scratch' (caller-owned, at least box2_tmp_bytes + 64) if given; otherwise a
 * temporary allocation released after the stream has drained. (hipMallocAsync /
 * hipFreeAsync would avoid the drain, but on this ROCm build kernels reading such an
 * allocation intermittently saw stale data: 1048 of 4000 randomized cases wrong,
 * none with plain hipMalloc -- tests/exp/fuzz_box2_gpu.py.) */
hipError_t
launch_single (JobDev &jd, hipStream_t s, void *scratch = nullptr)
{
    const KClass k = classify (jd);

    if (k != KC_BOX2)
        return launch_class (k, nullptr, 1, jd, jd.w_out, jd.n_rows, s);

    void *tmp = scratch;
    hipError_t e = hipSuccess;
    if (!tmp)
        e = hipMalloc (&tmp, box2_tmp_bytes (jd) + 64);
    if (e != hipSuccess)
        return e;
    jd.tmp = static_cast<uint8_t *> (tmp);
    e = launch_class (k, nullptr, 1, jd, jd.w_out, jd.n_rows, s);
    if (!scratch)
    {
        const hipError_t e2 = hipStreamSynchronize (s);
        (void) hipFree (tmp);
        if (e == hipSuccess)
            e = e2;
    }
    return e;
}

/* ---- per-device lanes: stream + grow-only staging buffers ---------------- */

struct Lane
{
    std::mutex mtx;
    bool ready = false;
    hipStream_t stream = nullptr;
    void *src_buf = nullptr;  size_t src_cap = 0;
    void *dst_buf = nullptr;  size_t dst_cap = 0;
    void *tmp_buf = nullptr;  size_t tmp_cap = 0;      /* intermediate image of the two-pass box filter */
};

constexpr int N_LANES = 4;
constexpr int MAX_DEVICES = 16;

struct DeviceState
{
    Lane lanes [N_LANES];
    std::atomic<unsigned> next { 0 };
};

DeviceState g_devices [MAX_DEVICES];

int
lane_prepare_tmp (Lane &ln, size_t need)
{
    if (need > ln.tmp_cap)
    {
        if (ln.tmp_buf)
            HIP_TRY (hipFree (ln.tmp_buf));
        ln.tmp_buf = nullptr; ln.tmp_cap = 0;
        HIP_TRY (hipMalloc (&ln.tmp_buf, need));
        ln.tmp_cap = need;
    }
    return 0;
}

int
lane_prepare (Lane &ln, size_t src_need, size_t dst_need)
{
    if (!ln.ready)
    {
        HIP_TRY (hipStreamCreateWithFlags (&ln.stream, hipStreamNonBlocking));
        ln.ready = true;
    }
    if (src_need > ln.src_cap)
    {
        if (ln.src_buf)
            HIP_TRY (hipFree (ln.src_buf));
        ln.src_buf = nullptr; ln.src_cap = 0;
        HIP_TRY (hipMalloc (&ln.src_buf, src_need));
        ln.src_cap = src_need;
    }
    if (dst_need > ln.dst_cap)
    {
        if (ln.dst_buf)
            HIP_TRY (hipFree (ln.dst_buf));
        ln.dst_buf = nullptr; ln.dst_cap = 0;
        HIP_TRY (hipMalloc (&ln.dst_buf, dst_need));
        ln.dst_cap = dst_need;
    }
    return 0;
}

} /* namespace */

struct SmolHipPlan
{
    SmolHipJob job;
    int device;
};

struct SmolHipBatch
{
    int device = 0;
    uint32_t n_images = 0;
    JobDev *jobs_dev = nullptr;            /* all jobs, grouped by launch */
    void *scratch = nullptr;               /* intermediate images of the two-pass box jobs */
    hipStream_t tstream = nullptr;         /* smol_hip_batch_time: stream and events, created once */
    hipEvent_t te0 = nullptr, te1 = nullptr;
    struct Group { KClass k; uint32_t first, count; JobDev rep; uint32_t max_w_out, max_rows, max_lds; };
    std::vector<Group> groups;
    uint64_t dominant_units = 0;
    const char *dominant_name = "";
    std::vector<JobDev> jobs_host;         /* same order as jobs_dev */
    hipGraph_t graph = nullptr;            /* optional: one kernel node per image */
    hipGraphExec_t graph_exec = nullptr;
    bool use_graph = false;
};

extern "C" {

int
smol_hip_device_count (void)
{
    int n = 0;
    if (hipGetDeviceCount (&n) != hipSuccess)
        return 0;
    return n;
}

int
smol_hip_plan_create (const SmolHipJob *job, int device, SmolHipPlan **plan_out)
{
    if (!job || !plan_out || device < 0 || device >= MAX_DEVICES)
        return SMOL_HIP_E_INVALID;
    if (job->pixel_type_in >= 10 || job->pixel_type_out >= 10)
        return SMOL_HIP_E_INVALID;
    if (device >= smol_hip_device_count ())
        return SMOL_HIP_E_NO_DEVICE;

    SmolHipPlan *p = new (std::nothrow) SmolHipPlan;
    if (!p)
        return SMOL_HIP_E_INVALID;
    p->job = *job;
    p->device = device;
    *plan_out = p;
    return 0;
}

void
smol_hip_plan_destroy (SmolHipPlan *plan)
{
    delete plan;
}

int
smol_hip_scale_rows_device (SmolHipPlan *plan,
                            const void *pixels_in_dev, uint32_t rowstride_in,
                            void *dest_dev, uint32_t rowstride_out,
                            uint32_t first_row, uint32_t n_rows,
                            void *hip_stream)
{
    if (!plan || !pixels_in_dev || !dest_dev)
        return SMOL_HIP_E_INVALID;
    if (n_rows == 0)
        return 0;
    if (first_row + n_rows > plan->job.height_out)
        return SMOL_HIP_E_INVALID;

    HIP_TRY (hipSetDevice (plan->device));

    JobDev jd;
    fill_jobdev (jd, plan->job, pixels_in_dev, rowstride_in, dest_dev, rowstride_out, first_row, n_rows);
    HIP_TRY (launch_single (jd, static_cast<hipStream_t> (hip_stream)));
    return 0;
}

int
smol_hip_scale_rows_host (SmolHipPlan *plan,
                          const void *pixels_in_host, uint32_t rowstride_in,
                          void *dest_host, uint32_t rowstride_out,
                          uint32_t first_row, uint32_t n_rows)
{
    if (!plan || !pixels_in_host || !dest_host)
        return SMOL_HIP_E_INVALID;
    if (n_rows == 0)
        return 0;

    const SmolHipJob &job = plan->job;
    uint32_t r0, r1;
    int rc = smol_hip_input_row_range (&job, first_row, n_rows, &r0, &r1);
    if (rc)
        return rc;

    HIP_TRY (hipSetDevice (plan->device));

    const TypeInfo &ti = k_types [job.pixel_type_in];
    const TypeInfo &to = k_types [job.pixel_type_out];
    const size_t in_row_bytes = (size_t) job.width_in * ti.bpp;
    const size_t out_row_bytes = (size_t) job.width_out * to.bpp;
    /* Device rows are re-pitched to 16-byte multiples whatever the host strides
     * are, so kernels see aligned rows even for odd host layouts (the reference
     * copies odd rows to an aligned temporary: smolscale-generic.c:1623-1634). */
    const size_t src_pitch = align_up (in_row_bytes, 16);
    const size_t dst_pitch = align_up (out_row_bytes, 16);
    const size_t n_in_rows = (size_t) r1 - r0 + 1;

    if (src_pitch > 0xffffffffu || dst_pitch > 0xffffffffu)
        return SMOL_HIP_E_INVALID;

    DeviceState &ds = g_devices [plan->device];
    Lane &ln = ds.lanes [ds.next.fetch_add (1) % N_LANES];
    std::lock_guard<std::mutex> guard (ln.mtx);

    rc = lane_prepare (ln, n_in_rows * src_pitch + 64, (size_t) n_rows * dst_pitch + 64);
    if (rc)
        return rc;

    const uint8_t *host_src = static_cast<const uint8_t *> (pixels_in_host) + (size_t) r0 * rowstride_in;
    HIP_TRY (hipMemcpy2DAsync (ln.src_buf, src_pitch, host_src, rowstride_in,
                               in_row_bytes, n_in_rows, hipMemcpyHostToDevice, ln.stream));

    /* Row 0 of the virtual source image lies r0 rows before the slab. */
    const uint8_t *virt_src = static_cast<const uint8_t *> (ln.src_buf) - (size_t) r0 * src_pitch;
    JobDev jd;
    fill_jobdev (jd, job, virt_src, (uint32_t) src_pitch, ln.dst_buf, (uint32_t) dst_pitch,
                 first_row, n_rows);
    {
        /* the two-pass box filter keeps its intermediate image in the lane */
        JobDev probe = jd;
        if (classify (probe) == KC_BOX2)
        {
            rc = lane_prepare_tmp (ln, box2_tmp_bytes (probe) + 64);
            if (rc)
                return rc;
            HIP_TRY (launch_single (jd, ln.stream, ln.tmp_buf));
        }
        else
            HIP_TRY (launch_single (jd, ln.stream));
    }
    HIP_TRY (hipMemcpy2DAsync (dest_host, rowstride_out, ln.dst_buf, dst_pitch,
                               out_row_bytes, n_rows, hipMemcpyDeviceToHost, ln.stream));
    HIP_TRY (hipStreamSynchronize (ln.stream));
    return 0;
}

/* ---- batches -------------------------------------------------------------- */

int
smol_hip_batch_create (const SmolHipImage *images, uint32_t n_images, int device,
                       SmolHipBatch **batch_out)
{
    if (!images || !n_images || !batch_out || device < 0 || device >= MAX_DEVICES)
        return SMOL_HIP_E_INVALID;
    if (device >= smol_hip_device_count ())
        return SMOL_HIP_E_NO_DEVICE;

    HIP_TRY (hipSetDevice (device));

    std::vector<JobDev> jobs (n_images);
    std::vector<KClass> cls (n_images);

    for (uint32_t i = 0; i < n_images; i++)
    {
        const SmolHipImage &im = images [i];

        if (im.job.pixel_type_in >= 10 || im.job.pixel_type_out >= 10 || !im.src || !im.dst
            || im.first_row + im.n_rows > im.job.height_out)
            return SMOL_HIP_E_INVALID;
        fill_jobdev (jobs [i], im.job, im.src, im.rowstride_in, im.dst, im.rowstride_out,
                     im.first_row, im.n_rows);
        cls [i] = classify (jobs [i]);
    }

    SmolHipBatch *b = new (std::nothrow) SmolHipBatch;
    if (!b)
        return SMOL_HIP_E_INVALID;
    b->device = device;
    b->n_images = n_images;

    /* Stable grouping: images that can share one launch (same kernel class and,
     * for k_pow2, same halvings) become one contiguous run of descriptors. */
    std::vector<JobDev> sorted;
    sorted.reserve (n_images);
    std::vector<bool> taken (n_images, false);

    for (uint32_t i = 0; i < n_images; i++)
    {
        if (taken [i])
            continue;

        SmolHipBatch::Group g;
        g.k = cls [i];
        g.first = (uint32_t) sorted.size ();
        g.count = 0;
        g.rep = jobs [i];
        g.max_w_out = 0;
        g.max_rows = 0;
        g.max_lds = 0;
        uint64_t units = 0;

        for (uint32_t j = i; j < n_images; j++)
        {
            if (taken [j] || cls [j] != g.k)
                continue;
            if (g.k == KC_POW2 && (jobs [j].halv_h != g.rep.halv_h || jobs [j].halv_v != g.rep.halv_v))
                continue;
            if ((g.k == KC_TILE || g.k == KC_WTILE) && (jobs [j].tile_rows != g.rep.tile_rows || jobs [j].tile_cpl != g.rep.tile_cpl
                                  || (jobs [j].tile_variant >> 2) != (g.rep.tile_variant >> 2)
                                  || jobs [j].tile_lpc != g.rep.tile_lpc || jobs [j].tile_vcopy != g.rep.tile_vcopy))
                continue;
            if ((g.k == KC_WAVEPX || g.k == KC_WAVEPX_WIDE) && jobs [j].w_out != g.rep.w_out)
                continue;
            if (g.k == KC_POW2X && (jobs [j].halv_h != g.rep.halv_h || jobs [j].px_kind != g.rep.px_kind))
                continue;
            if (g.k == KC_DOWN && (jobs [j].halv_h != g.rep.halv_h || jobs [j].tile_cpl != g.rep.tile_cpl || jobs [j].down_mode != g.rep.down_mode))
                continue;
            if (g.k == KC_UP && (jobs [j].tile_aligned4 != g.rep.tile_aligned4 || jobs [j].up_wide != g.rep.up_wide
                                 || (jobs [j].tile_variant & 3u) != (g.rep.tile_variant & 3u)))
                continue;
            if (g.k == KC_BOX2 && jobs [j].wide != g.rep.wide)
                continue;
            if (g.count == 65535)
                break;
            taken [j] = true;
            sorted.push_back (jobs [j]);
            g.count++;
            if (g.k == KC_BOX2)
            {
                /* max_w_out / max_rows / max_lds = pass-1 waves / pass-2 waves / pass-1 LDS bytes */
                if (box2_h_tiles (jobs [j]) > g.max_w_out) g.max_w_out = box2_h_tiles (jobs [j]);
                if (box2_v_tiles (jobs [j]) > g.max_rows) g.max_rows = box2_v_tiles (jobs [j]);
                if (box2_h_lds_bytes (jobs [j]) > g.max_lds) g.max_lds = box2_h_lds_bytes (jobs [j]);
                units += (uint64_t) jobs [j].w_out * jobs [j].n_rows;
                continue;
            }
            if (g.k == KC_UP)
            {
                if (up_lds_bytes (jobs [j]) > g.max_w_out) g.max_w_out = up_lds_bytes (jobs [j]);
            }
            else if (g.k == KC_DOWN)
            {
                if (down_lds_bytes (jobs [j]) > g.max_w_out) g.max_w_out = down_lds_bytes (jobs [j]);
            }
            else if (jobs [j].w_out > g.max_w_out) g.max_w_out = jobs [j].w_out;
            if (jobs [j].n_rows > g.max_rows) g.max_rows = jobs [j].n_rows;
            if ((g.k == KC_TILE || g.k == KC_WTILE) && tile_lds_bytes (jobs [j]) > g.max_lds) g.max_lds = tile_lds_bytes (jobs [j]);
            if (g.k == KC_POW2X && pow2x_blocks (jobs [j]) > g.max_lds) g.max_lds = pow2x_blocks (jobs [j]);
            if (g.k == KC_UP && up_blocks (jobs [j]) > g.max_lds) g.max_lds = up_blocks (jobs [j]);
            if (g.k == KC_DOWN && down_blocks (jobs [j]) > g.max_lds) g.max_lds = down_blocks (jobs [j]);
            units += (uint64_t) jobs [j].w_out * jobs [j].n_rows;
        }

        if (units > b->dominant_units)
        {
            b->dominant_units = units;
            b->dominant_name = kclass_name (g.k, g.rep);
        }
        if (g.max_rows > 0)
            b->groups.push_back (g);
    }

    /* intermediate images of the two-pass box jobs: one allocation, carved up per job */
    hipError_t e = hipSuccess;
    {
        size_t total = 0;
        for (const JobDev &jd : sorted)
            if (box2_is (jd))
                total += align_up (box2_tmp_bytes (jd) + 64, 256);
        if (total)
        {
            e = hipMalloc (&b->scratch, total);
            size_t off = 0;
            if (e == hipSuccess)
                for (JobDev &jd : sorted)
                    if (box2_is (jd))
                    {
                        jd.tmp = static_cast<uint8_t *> (b->scratch) + off;
                        off += align_up (box2_tmp_bytes (jd) + 64, 256);
                    }
        }
    }
    if (e == hipSuccess)
        e = hipMalloc (reinterpret_cast<void **> (&b->jobs_dev), sorted.size () * sizeof (JobDev));
    if (e == hipSuccess)
        e = hipMemcpy (b->jobs_dev, sorted.data (), sorted.size () * sizeof (JobDev), hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
        if (b->jobs_dev)
            (void) hipFree (b->jobs_dev);
        if (b->scratch)
            (void) hipFree (b->scratch);
        delete b;
        return -(int) e;
    }

    b->jobs_host = sorted;
    *batch_out = b;
    return 0;
}

/* Alternative submission form for the same batch: every image becomes its own
 * kernel node (descriptor by value) of a hipGraph captured once and replayed by
 * smol_hip_batch_launch(). Exists to measure BASELINE config 5's "hipGraph-captured
 * per-image launches" against the default one-launch-per-kernel-class form. */
int
smol_hip_batch_use_graph (SmolHipBatch *batch, int enable)
{
    if (!batch)
        return SMOL_HIP_E_INVALID;
    if (!enable)
    {
        batch->use_graph = false;
        return 0;
    }
    HIP_TRY (hipSetDevice (batch->device));
    if (!batch->graph_exec)
    {
        hipStream_t cs;
        HIP_TRY (hipStreamCreateWithFlags (&cs, hipStreamNonBlocking));
        HIP_TRY (hipStreamBeginCapture (cs, hipStreamCaptureModeThreadLocal));
        hipError_t e = hipSuccess;
        for (JobDev &jd : batch->jobs_host)
        {
            if (jd.n_rows == 0)
                continue;
            /* (two-pass jobs keep the intermediate image the batch allocated for them) */
            uint8_t *tmp_keep = jd.tmp;
            const KClass kc = classify (jd);
            jd.tmp = tmp_keep;
            e = launch_class (kc, nullptr, 1, jd, jd.w_out, jd.n_rows, cs);
            if (e != hipSuccess)
                break;
        }
        hipError_t e2 = hipStreamEndCapture (cs, &batch->graph);
        (void) hipStreamDestroy (cs);
        if (e != hipSuccess)
            return -(int) e;
        if (e2 != hipSuccess)
            return -(int) e2;
        HIP_TRY (hipGraphInstantiate (&batch->graph_exec, batch->graph, nullptr, nullptr, 0));
    }
    batch->use_graph = true;
    return 0;
}

int
smol_hip_batch_launch (SmolHipBatch *batch, void *hip_stream)
{
    if (!batch)
        return SMOL_HIP_E_INVALID;

    hipStream_t s = static_cast<hipStream_t> (hip_stream);

    HIP_TRY (hipSetDevice (batch->device));
    if (batch->use_graph)
    {
        HIP_TRY (hipGraphLaunch (batch->graph_exec, s));
        return 0;
    }
    for (const SmolHipBatch::Group &g : batch->groups)
        HIP_TRY (launch_class (g.k, batch->jobs_dev + g.first, g.count, g.rep, g.max_w_out, g.max_rows, s, g.max_lds));
    return 0;
}

int
smol_hip_batch_info (const SmolHipBatch *batch, uint32_t *n_launches, const char **kernel_name)
{
    if (!batch)
        return SMOL_HIP_E_INVALID;
    if (n_launches)
        *n_launches = (uint32_t) batch->groups.size ();
    if (kernel_name)
        *kernel_name = batch->dominant_name;
    return 0;
}

void
smol_hip_batch_destroy (SmolHipBatch *batch)
{
    if (!batch)
        return;
    if (hipSetDevice (batch->device) == hipSuccess)
    {
        if (batch->jobs_dev)
            (void) hipFree (batch->jobs_dev);
        if (batch->scratch)
            (void) hipFree (batch->scratch);
        if (batch->tstream)
        {
            (void) hipEventDestroy (batch->te0);
            (void) hipEventDestroy (batch->te1);
            (void) hipStreamDestroy (batch->tstream);
        }
        if (batch->graph_exec)
            (void) hipGraphExecDestroy (batch->graph_exec);
        if (batch->graph)
            (void) hipGraphDestroy (batch->graph);
    }
    delete batch;
}

int
smol_hip_batch_time (SmolHipBatch *batch, uint32_t warmup, uint32_t iters, float *ms_total)
{
    if (!batch || !ms_total || !iters)
        return SMOL_HIP_E_INVALID;

    HIP_TRY (hipSetDevice (batch->device));

    /* the timing stream and its events live with the batch: creating a stream costs a few
     * hundred microseconds, which a caller timing K launches would see as a fixed overhead */
    if (!batch->tstream)
    {
        HIP_TRY (hipStreamCreateWithFlags (&batch->tstream, hipStreamNonBlocking));
        HIP_TRY (hipEventCreate (&batch->te0));
        HIP_TRY (hipEventCreate (&batch->te1));
    }
    const hipStream_t s = batch->tstream;
    const hipEvent_t e0 = batch->te0, e1 = batch->te1;

    int rc = 0;
    for (uint32_t i = 0; i < warmup && !rc; i++)
        rc = smol_hip_batch_launch (batch, s);
    if (!rc)
    {
        HIP_TRY (hipStreamSynchronize (s));
        HIP_TRY (hipEventRecord (e0, s));
        for (uint32_t i = 0; i < iters && !rc; i++)
            rc = smol_hip_batch_launch (batch, s);
        HIP_TRY (hipEventRecord (e1, s));
        HIP_TRY (hipEventSynchronize (e1));
        HIP_TRY (hipEventElapsedTime (ms_total, e0, e1));
    }

    return rc;
}

/* ---- memory helpers ---------------------------------------------------------- */

int
smol_hip_malloc (int device, uint64_t n_bytes, void **dev_ptr_out)
{
    if (!dev_ptr_out)
        return SMOL_HIP_E_INVALID;
    HIP_TRY (hipSetDevice (device));
    HIP_TRY (hipMalloc (dev_ptr_out, n_bytes));
    return 0;
}

int
smol_hip_free (int device, void *dev_ptr)
{
    HIP_TRY (hipSetDevice (device));
    HIP_TRY (hipFree (dev_ptr));
    return 0;
}

int
smol_hip_memcpy_h2d (int device, void *dst_dev, const void *src_host, uint64_t n_bytes)
{
    HIP_TRY (hipSetDevice (device));
    HIP_TRY (hipMemcpy (dst_dev, src_host, n_bytes, hipMemcpyHostToDevice));
    return 0;
}

int
smol_hip_memcpy_d2h (int device, void *dst_host, const void *src_dev, uint64_t n_bytes)
{
    HIP_TRY (hipSetDevice (device));
    HIP_TRY (hipMemcpy (dst_host, src_dev, n_bytes, hipMemcpyDeviceToHost));
    return 0;
}

int
smol_hip_memset (int device, void *dst_dev, int value, uint64_t n_bytes)
{
    HIP_TRY (hipSetDevice (device));
    HIP_TRY (hipMemset (dst_dev, value, n_bytes));
    return 0;
}

int
smol_hip_device_synchronize (int device)
{
    HIP_TRY (hipSetDevice (device));
    HIP_TRY (hipDeviceSynchronize ());
    return 0;
}

} /* extern "C" */
