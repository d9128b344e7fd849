/* smol_rt_batch.hip -- batches: many images (device pointers) in as few launches as possible, the same as a
 * hipGraph with one kernel node per image, a batch dealt out over several devices, and the timing helpers
 * bench.py uses. */

#include "smol_rt.h"

#include <algorithm>
#include <new>
#include <vector>

using namespace smolhip::dispatch;
using namespace smolrt;

struct SmolHipBatch
{
    int device = 0;
    uint32_t n_images = 0;
    JobDev *jobs_dev = nullptr;            /* all jobs, grouped by launch */
    void *scratch = nullptr;               /* intermediate images of the two-pass box jobs */
    hipStream_t tstream = nullptr;         /* smol_hip_batch_time: stream and events, created once */
    hipEvent_t te0 = nullptr, te1 = nullptr;
    std::vector<hipEvent_t> tev;           /* smol_hip_batch_time_each */
    struct Group { KClass k; uint32_t first, count; JobDev rep; uint32_t max_w_out, max_rows, max_lds; };
    std::vector<Group> groups;
    uint64_t dominant_units = 0;
    const char *dominant_name = "";
    std::vector<JobDev> jobs_host;         /* same order as jobs_dev */
    hipGraph_t graph = nullptr;            /* optional: one kernel node per image */
    hipGraphExec_t graph_exec = nullptr;
    bool use_graph = false;
};

struct SmolHipBatchMulti
{
    std::vector<SmolHipBatch *> parts;     /* one per listed device (entries with no image: nullptr) */
};

extern "C" {

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

    std::vector<JobDev> jobs;
    std::vector<KClass> cls;
    jobs.reserve (n_images);
    cls.reserve (n_images);

    /* what the launch reads in all: the box filter's choice between its single-pass and two-pass kernels goes by it */
    uint64_t src_bytes = 0;
    for (uint32_t i = 0; i < n_images; i++)
        if (images [i].job.pixel_type_in < 10 && images [i].job.height_out)
            src_bytes += (uint64_t) images [i].job.width_in * images [i].job.height_in * (images [i].job.pixel_type_in >= 8 ? 3u : 4u)
                         * images [i].n_rows / images [i].job.height_out;
    const uint16_t launch_src_mb = (uint16_t) ((src_bytes >> 20) > 65535u ? 65535u : (src_bytes >> 20));

    for (uint32_t i = 0; i < n_images; i++)
    {
        const SmolHipImage &im = images [i];

        if (im.job.pixel_type_in >= 10 || im.job.pixel_type_out >= 10)
            return SMOL_HIP_E_INVALID;
        if (im.n_rows == 0)
            continue;                   /* an empty strip (more ranks than rows): nothing to launch */
        if (!im.src || !im.dst || !rows_ok (im.first_row, im.n_rows, im.job.height_out))
            return SMOL_HIP_E_INVALID;
        JobDev jd;
        fill_jobdev (jd, im.job, im.src, im.rowstride_in, im.dst, im.rowstride_out,
                     im.first_row, im.n_rows);
        jd.launch_src_mb = launch_src_mb;
        cls.push_back (classify (jd));
        jobs.push_back (jd);
    }
    n_images = (uint32_t) jobs.size ();

    /* Exact 2:1 / 4:1 / 8:1 jobs of several flavours (k_pow2 and the input kinds of k_pow2x) with the same
     * halvings travel in ONE launch in the caller's order (k_pow2x_mixed_batch) instead of one launch per flavour. */
    for (uint32_t h = 0; h < 3; h++)
    {
        uint32_t seen = 0;                          /* bit per flavour: 0 = k_pow2, 1 + kind = k_pow2x */
        for (uint32_t i = 0; i < n_images; i++)
        {
            if (jobs [i].halv_h != h || jobs [i].halv_v != h)
                continue;
            if (cls [i] == KC_POW2)
                seen |= 1u;
            else if (cls [i] == KC_POW2X)
                seen |= 2u << jobs [i].px_kind;
        }
        if ((seen & (seen - 1)) == 0 || smolhip::tune_env ("SMOL_BATCH_NO_MIXED_POW2X"))
            continue;                               /* at most one flavour */
        for (uint32_t i = 0; i < n_images; i++)
        {
            if (jobs [i].halv_h != h || jobs [i].halv_v != h)
                continue;
            if (cls [i] == KC_POW2)
            {
                JobDev t = jobs [i];
                if (!pow2x_setup (t))
                    continue;                       /* stays a k_pow2 job of its own group */
                jobs [i] = t;
                cls [i] = KC_POW2X;
            }
            if (cls [i] == KC_POW2X)
                jobs [i].px_mixed = 1;
        }
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
                                  || jobs [j].tile_lpc != g.rep.tile_lpc || jobs [j].tile_vm != g.rep.tile_vm))
                continue;
            if ((g.k == KC_WAVEPX || g.k == KC_WAVEPX_WIDE) && jobs [j].w_out != g.rep.w_out)
                continue;
            if (g.k == KC_POW2X && (jobs [j].halv_h != g.rep.halv_h || jobs [j].px_mixed != g.rep.px_mixed
                                    || (!g.rep.px_mixed && jobs [j].px_kind != g.rep.px_kind)))
                continue;
            if (g.k == KC_DOWN && (jobs [j].halv_h != g.rep.halv_h || jobs [j].bpp_in != g.rep.bpp_in || jobs [j].p16 != g.rep.p16
                                   || lind_narrow (jobs [j]) != lind_narrow (g.rep) || lind_nt (jobs [j]) != lind_nt (g.rep) || lind_hv0 (jobs [j]) != lind_hv0 (g.rep)))
                continue;
            if (g.k == KC_UP && (jobs [j].tile_aligned4 != g.rep.tile_aligned4 || jobs [j].up_wide != g.rep.up_wide
                                 || (jobs [j].tile_variant & 3u) != (g.rep.tile_variant & 3u)))
                continue;
            if (g.k == KC_BOX2 && (jobs [j].wide != g.rep.wide || jobs [j].box2_ygroup != g.rep.box2_ygroup || jobs [j].box2 != g.rep.box2
                                   || (jobs [j].tile_lpc >= 16) != (g.rep.tile_lpc >= 16)
                                   /* k_linh / k_linv take their KIND (11-bit, 24-bit rows, 19-bit, premul16) from the group's
                                    * representative: everything lin_pass_kind () reads is part of the key */
                                   || (g.rep.box2 >= 2 && (jobs [j].filter_h != g.rep.filter_h || jobs [j].filter_v != g.rep.filter_v
                                                           || jobs [j].halv_h != g.rep.halv_h || jobs [j].bpp_in != g.rep.bpp_in
                                                           || jobs [j].p16 != g.rep.p16 || jobs [j].linear != g.rep.linear))))
                continue;
            if (g.k == KC_HSPAN && (jobs [j].bpp_in != g.rep.bpp_in || jobs [j].halv_h != g.rep.halv_h))
                continue;
            if (g.count == 65535)
                break;
            taken [j] = true;
            sorted.push_back (jobs [j]);
            g.count++;
            if (g.k == KC_BOX2)
            {
                /* max_w_out / max_rows / max_lds = pass-1 waves / pass-2 waves / pass-1 LDS bytes */
                g.max_w_out = box2_grid_max (g.max_w_out, box2_h_grid (jobs [j]));
                if (box2_v_tiles (jobs [j]) > g.max_rows) g.max_rows = box2_v_tiles (jobs [j]);
                if (box2_h_lds_bytes (jobs [j]) > g.max_lds) g.max_lds = box2_h_lds_bytes (jobs [j]);
                units += (uint64_t) jobs [j].w_out * jobs [j].n_rows;
                continue;
            }
            if (g.k == KC_POW2X)
            {
                if (pow2x_tiles_x (jobs [j]) > g.max_w_out) g.max_w_out = pow2x_tiles_x (jobs [j]);
                if (pow2x_tiles_y (jobs [j]) > g.max_rows) g.max_rows = pow2x_tiles_y (jobs [j]);
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
            if (g.k == KC_UP && up_blocks (jobs [j]) > g.max_lds) g.max_lds = up_blocks (jobs [j]);
            if (g.k == KC_DOWN && down_blocks (jobs [j]) > g.max_lds) g.max_lds = down_blocks (jobs [j]);
            if (g.k == KC_HSPAN && hspan_blocks (jobs [j]) > g.max_lds) g.max_lds = hspan_blocks (jobs [j]);
            units += (uint64_t) jobs [j].w_out * jobs [j].n_rows;
        }

        /* experiment (tuning builds): the images of a mixed-flavour launch ordered by flavour */
        if (g.k == KC_POW2X && g.rep.px_mixed && smolhip::tune_env ("SMOL_BATCH_SORT_MIXED"))
            std::stable_sort (sorted.begin () + g.first, sorted.end (),
                              [] (const JobDev &a, const JobDev &b2) { return a.px_kind < b2.px_kind; });
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
    if (e == hipSuccess && !sorted.empty ())
        e = hipMalloc (reinterpret_cast<void **> (&b->jobs_dev), sorted.size () * sizeof (JobDev));
    if (e == hipSuccess && !sorted.empty ())
    {
        /* (through the bounce ring: the descriptor table of 10 000 thumbnails is 2 MB of pageable heap) */
        const int trc = transfer_h2d (device, b->jobs_dev, sorted.data (), sorted.size () * sizeof (JobDev));
        if (trc)
            e = trc > -10000 ? (hipError_t) -trc : hipErrorUnknown;
    }
    if (e != hipSuccess)
    {
        if (b->jobs_dev)
            (void) hipFree (b->jobs_dev);
        if (b->scratch)
            (void) hipFree (b->scratch);
        delete b;
        return -(int) e;
    }

    /* the representative of a group is its first descriptor as finally laid out (scratch pointers included) */
    for (SmolHipBatch::Group &g : b->groups)
        g.rep = sorted [g.first];
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
        /* The images are independent: captured on GRAPH_BRANCHES forked streams, so the graph is that many
         * parallel chains of kernel nodes instead of one chain of n_images (10 000 thumbnails: 28.9 ms as one
         * chain, 2.9 us per node). */
        constexpr int GRAPH_BRANCHES = 8;
        hipStream_t cs, br [GRAPH_BRANCHES] = {};
        hipEvent_t ev_fork = nullptr, ev_join [GRAPH_BRANCHES] = {};
        HIP_TRY (hipStreamCreateWithFlags (&cs, hipStreamNonBlocking));
        hipError_t e = hipEventCreateWithFlags (&ev_fork, hipEventDisableTiming);
        for (int i = 0; i < GRAPH_BRANCHES && e == hipSuccess; i++)
        {
            e = hipStreamCreateWithFlags (&br [i], hipStreamNonBlocking);
            if (e == hipSuccess)
                e = hipEventCreateWithFlags (&ev_join [i], hipEventDisableTiming);
        }
        hipError_t e2 = hipSuccess;
        if (e == hipSuccess)
        {
            e = hipStreamBeginCapture (cs, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess)
            {
                e = hipEventRecord (ev_fork, cs);
                for (int i = 0; i < GRAPH_BRANCHES && e == hipSuccess; i++)
                    e = hipStreamWaitEvent (br [i], ev_fork, 0);
                size_t k = 0;
                for (JobDev &jd : batch->jobs_host)
                {
                    if (e != hipSuccess)
                        break;
                    if (jd.n_rows == 0)
                        continue;
                    /* (two-pass jobs keep the intermediate image the batch allocated for them) */
                    uint8_t *tmp_keep = jd.tmp;
                    const KClass kc = classify (jd);
                    jd.tmp = tmp_keep;
                    e = launch_class (kc, nullptr, 1, jd, jd.w_out, jd.n_rows, br [k++ % GRAPH_BRANCHES]);
                }
                for (int i = 0; i < GRAPH_BRANCHES; i++)
                {
                    hipError_t ej = hipEventRecord (ev_join [i], br [i]);
                    if (ej == hipSuccess)
                        ej = hipStreamWaitEvent (cs, ev_join [i], 0);
                    if (e == hipSuccess)
                        e = ej;
                }
                e2 = hipStreamEndCapture (cs, &batch->graph);
            }
        }
        for (int i = 0; i < GRAPH_BRANCHES; i++)
        {
            if (ev_join [i]) (void) hipEventDestroy (ev_join [i]);
            if (br [i]) (void) hipStreamDestroy (br [i]);
        }
        if (ev_fork) (void) hipEventDestroy (ev_fork);
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
    /* a group of one travels in the kernel arguments (no descriptor fetch at the head of every workgroup) */
    for (const SmolHipBatch::Group &g : batch->groups)
        HIP_TRY (launch_class (g.k, g.count == 1 ? nullptr : batch->jobs_dev + g.first, g.count, g.rep,
                               g.max_w_out, g.max_rows, s, g.max_lds));
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
        for (hipEvent_t ev : batch->tev)
            (void) hipEventDestroy (ev);
        if (batch->graph_exec)
            (void) hipGraphExecDestroy (batch->graph_exec);
        if (batch->graph)
            (void) hipGraphDestroy (batch->graph);
    }
    delete batch;
}

namespace {

/* the timing stream and its events live with the batch: creating a stream costs a few
 * hundred microseconds, which a caller timing K launches would see as a fixed overhead */
int
batch_timing_setup (SmolHipBatch *batch)
{
    HIP_TRY (hipSetDevice (batch->device));
    if (!batch->tstream)
    {
        HIP_TRY (hipStreamCreateWithFlags (&batch->tstream, hipStreamNonBlocking));
        HIP_TRY (hipEventCreate (&batch->te0));
        HIP_TRY (hipEventCreate (&batch->te1));
    }
    return 0;
}

/* warm-up launches, wait, e0, `iters' launches, e1 -- all asynchronous after the wait */
int
batch_time_issue (SmolHipBatch *batch, uint32_t warmup, uint32_t iters)
{
    int rc = batch_timing_setup (batch);
    const hipStream_t s = batch->tstream;

    for (uint32_t i = 0; i < warmup && !rc; i++)
        rc = smol_hip_batch_launch (batch, s);
    if (rc)
        return rc;
    HIP_TRY (hipStreamSynchronize (s));
    HIP_TRY (hipEventRecord (batch->te0, s));
    for (uint32_t i = 0; i < iters && !rc; i++)
        rc = smol_hip_batch_launch (batch, s);
    HIP_TRY (hipEventRecord (batch->te1, s));
    return rc;
}

int
batch_time_collect (SmolHipBatch *batch, float *ms_total)
{
    HIP_TRY (hipSetDevice (batch->device));
    HIP_TRY (hipEventSynchronize (batch->te1));
    HIP_TRY (hipEventElapsedTime (ms_total, batch->te0, batch->te1));
    return 0;
}

} /* namespace */

int
smol_hip_batch_time (SmolHipBatch *batch, uint32_t warmup, uint32_t iters, float *ms_total)
{
    if (!batch || !ms_total || !iters)
        return SMOL_HIP_E_INVALID;

    int rc = batch_time_issue (batch, warmup, iters);
    const int rc2 = batch_time_collect (batch, ms_total);
    return rc ? rc : rc2;
}

int
smol_hip_batch_time_each (SmolHipBatch *batch, uint32_t warmup, uint32_t iters, float *ms_each)
{
    if (!batch || !ms_each || !iters)
        return SMOL_HIP_E_INVALID;

    int rc = batch_timing_setup (batch);
    if (rc)
        return rc;
    while (batch->tev.size () < (size_t) iters + 1)
    {
        hipEvent_t ev;
        HIP_TRY (hipEventCreate (&ev));
        batch->tev.push_back (ev);
    }
    const hipStream_t s = batch->tstream;

    for (uint32_t i = 0; i < warmup && !rc; i++)
        rc = smol_hip_batch_launch (batch, s);
    if (rc)
        return rc;
    HIP_TRY (hipStreamSynchronize (s));
    HIP_TRY (hipEventRecord (batch->tev [0], s));
    for (uint32_t i = 0; i < iters && !rc; i++)
    {
        rc = smol_hip_batch_launch (batch, s);
        HIP_TRY (hipEventRecord (batch->tev [i + 1], s));
    }
    HIP_TRY (hipEventSynchronize (batch->tev [iters]));
    for (uint32_t i = 0; i < iters; i++)
        HIP_TRY (hipEventElapsedTime (&ms_each [i], batch->tev [i], batch->tev [i + 1]));
    return rc;
}

int
smol_hip_batch_time_rotating (SmolHipBatch *const *batches, uint32_t n_batches, uint32_t warmup, uint32_t iters,
                              float *ms_total, float *ms_each)
{
    if (!batches || !n_batches || !ms_total || !iters)
        return SMOL_HIP_E_INVALID;
    for (uint32_t k = 0; k < n_batches; k++)
        if (!batches [k] || batches [k]->device != batches [0]->device)
            return SMOL_HIP_E_INVALID;

    SmolHipBatch *lead = batches [0];               /* its stream and events time the whole rotation */
    int rc = batch_timing_setup (lead);
    if (rc)
        return rc;
    while (ms_each && lead->tev.size () < (size_t) iters + 1)
    {
        hipEvent_t ev;
        HIP_TRY (hipEventCreate (&ev));
        lead->tev.push_back (ev);
    }
    const hipStream_t s = lead->tstream;

    for (uint32_t i = 0; i < warmup && !rc; i++)
        rc = smol_hip_batch_launch (batches [i % n_batches], s);
    if (rc)
        return rc;
    HIP_TRY (hipStreamSynchronize (s));
    HIP_TRY (hipEventRecord (lead->te0, s));
    if (ms_each)
        HIP_TRY (hipEventRecord (lead->tev [0], s));
    for (uint32_t i = 0; i < iters && !rc; i++)
    {
        rc = smol_hip_batch_launch (batches [(warmup + i) % n_batches], s);
        if (ms_each)
            HIP_TRY (hipEventRecord (lead->tev [i + 1], s));
    }
    HIP_TRY (hipEventRecord (lead->te1, s));
    HIP_TRY (hipEventSynchronize (lead->te1));
    HIP_TRY (hipEventElapsedTime (ms_total, lead->te0, lead->te1));
    for (uint32_t i = 0; ms_each && i < iters; i++)
        HIP_TRY (hipEventElapsedTime (&ms_each [i], lead->tev [i], lead->tev [i + 1]));
    return rc;
}

/* ---- a batch over several devices (image i -> devices [i % n_devices]) ------------- */

int
smol_hip_batch_create_multi (const SmolHipImage *images, uint32_t n_images,
                             const int *devices, uint32_t n_devices,
                             SmolHipBatchMulti **batch_out)
{
    if (!images || !n_images || !devices || !n_devices || n_devices > (uint32_t) MAX_DEVICES || !batch_out)
        return SMOL_HIP_E_INVALID;

    SmolHipBatchMulti *m = new (std::nothrow) SmolHipBatchMulti;
    if (!m)
        return SMOL_HIP_E_INVALID;
    m->parts.assign (n_devices, nullptr);

    std::vector<SmolHipImage> mine;
    for (uint32_t d = 0; d < n_devices; d++)
    {
        mine.clear ();
        for (uint32_t i = d; i < n_images; i += n_devices)
            mine.push_back (images [i]);
        if (mine.empty ())
            continue;
        const int rc = smol_hip_batch_create (mine.data (), (uint32_t) mine.size (), devices [d], &m->parts [d]);
        if (rc)
        {
            smol_hip_batch_multi_destroy (m);
            return rc;
        }
        /* every part runs on a stream of its own, the one its timing helper uses too */
        const int rc2 = batch_timing_setup (m->parts [d]);
        if (rc2)
        {
            smol_hip_batch_multi_destroy (m);
            return rc2;
        }
    }
    *batch_out = m;
    return 0;
}

int
smol_hip_batch_multi_launch (SmolHipBatchMulti *batch)
{
    if (!batch)
        return SMOL_HIP_E_INVALID;
    int rc = 0;
    for (SmolHipBatch *b : batch->parts)
        if (b)
        {
            const int r = smol_hip_batch_launch (b, b->tstream);
            if (r && !rc)
                rc = r;
        }
    return rc;
}

int
smol_hip_batch_multi_sync (SmolHipBatchMulti *batch)
{
    if (!batch)
        return SMOL_HIP_E_INVALID;
    for (SmolHipBatch *b : batch->parts)
        if (b)
        {
            HIP_TRY (hipSetDevice (b->device));
            HIP_TRY (hipStreamSynchronize (b->tstream));
        }
    return 0;
}

int
smol_hip_batch_multi_time (SmolHipBatchMulti *batch, uint32_t warmup, uint32_t iters, float *ms_per_device)
{
    if (!batch || !ms_per_device || !iters)
        return SMOL_HIP_E_INVALID;

    int rc = 0;
    /* every device first gets its whole queue (warm-up, e0, launches, e1), then all are collected:
     * the devices run concurrently, the host thread only feeds them */
    for (SmolHipBatch *b : batch->parts)
        if (b)
        {
            const int r = batch_time_issue (b, warmup, iters);
            if (r && !rc)
                rc = r;
        }
    for (size_t d = 0; d < batch->parts.size (); d++)
    {
        ms_per_device [d] = 0.0f;
        if (batch->parts [d])
        {
            const int r = batch_time_collect (batch->parts [d], &ms_per_device [d]);
            if (r && !rc)
                rc = r;
        }
    }
    return rc;
}

void
smol_hip_batch_multi_destroy (SmolHipBatchMulti *batch)
{
    if (!batch)
        return;
    for (SmolHipBatch *b : batch->parts)
        smol_hip_batch_destroy (b);
    delete batch;
}

} /* extern "C" */
