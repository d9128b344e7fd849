/* smol_rt_plan.hip -- plans (a job bound to a device), the ring of intermediate images the two-pass box filter
 * keeps per plan, and the device-pointer entry. */

#include "smol_rt.h"

#include <new>

using namespace smolhip::dispatch;
using namespace smolrt;

typedef SmolBoxSlot BoxSlot;
typedef SmolBoxUse BoxUse;
constexpr int BOX_SLOTS = SMOL_BOX_SLOTS;

namespace {

/* could this plan's rows go through an intermediate image (two-pass box filter, or the two-pass road of the
 * gamma-correct jobs with mixed axes)? Decided as the dispatcher would for aligned rows -- alignment can only take a
 * job OFF these roads -- and independent of the row range. */
bool
plan_may_need_scratch (const SmolHipPlan *plan)
{
    const SmolHipJob &j = plan->job;
    JobDev probe;
    fill_jobdev (probe, j, reinterpret_cast<const void *> (uintptr_t (256)), j.width_in * k_types [j.pixel_type_in].bpp,
                 reinterpret_cast<void *> (uintptr_t (256)), j.width_out * k_types [j.pixel_type_out].bpp, 0, j.height_out);
    /* 64bpp-class box x box takes k_tile or the two passes by the bytes a CALL reads (box2_setup): the whole image may
     * say k_tile where a later strip says two passes, and that strip must not find itself allocating under stream
     * capture -- so ask "could any row range take the two passes?" */
    probe.box_force = 2;
    return classify (probe) == KC_BOX2;
}

size_t
plan_box_bytes (const SmolHipPlan *plan)
{
    /* (16 bytes per pixel between the two k_lin* passes of a premul16 job, 8 everywhere else) */
    return (size_t) plan->job.height_in * plan->job.width_out * (plan->job.premul16 ? 16u : 8u) + 64;
}

int
box_slot_alloc (SmolHipPlan *plan, BoxSlot &sl)
{
    if (!sl.buf)
        HIP_TRY (hipMalloc (&sl.buf, plan_box_bytes (plan)));
    for (BoxUse &u : sl.use)
        if (!u.done)
            HIP_TRY (hipEventCreateWithFlags (&u.done, hipEventDisableTiming));
    return 0;
}

/* Stream order protects a slot only between calls on the SAME queue. hipStreamPerThread is one handle value that
 * stands for a different stream in every thread: two calls that both carry it may or may not share
 * a queue, so it never counts as "the same stream" -- such calls always go through the events. */
inline bool
same_queue (hipStream_t a, hipStream_t b)
{
    return a == b && a != hipStreamPerThread;
}

/* no call of another stream is still using the slot */
bool
box_slot_free_for (BoxSlot &sl, hipStream_t stream)
{
    bool ok = true;
    for (BoxUse &u : sl.use)
    {
        if (u.pending && !same_queue (u.stream, stream) && hipEventQuery (u.done) == hipSuccess)
            u.pending = false;
        if (u.pending && !same_queue (u.stream, stream))
            ok = false;
    }
    (void) hipGetLastError ();          /* hipErrorNotReady from the queries is not an error */
    return ok;
}

/* Picks the intermediate image for a call on `stream' reading image `src', makes the stream wait where it has to,
 * and returns the use entry the caller records behind its kernels. Under stream capture slot 0 is used as it
 * stands (no events: replays of a captured call must not overlap other work on the same plan -- smol_hip.h).
 * Called with the plan's mutex held. */
int
plan_box_acquire (SmolHipPlan *plan, const void *src, hipStream_t stream, bool capturing,
                  BoxSlot **slot_out, BoxUse **use_out)
{
    BoxSlot *pick = nullptr;

    *use_out = nullptr;
    if (capturing)
    {
        int rc = box_slot_alloc (plan, plan->box [0]);
        if (rc)
            return rc;
        *slot_out = &plan->box [0];
        return 0;
    }
    for (BoxSlot &sl : plan->box)
        if (sl.buf && sl.src == src)
            pick = &sl;
    if (!pick)
        for (BoxSlot &sl : plan->box)
            if (sl.buf && box_slot_free_for (sl, stream))
            {
                pick = &sl;
                break;
            }
    if (!pick)
        for (BoxSlot &sl : plan->box)
            if (!sl.buf)
            {
                if (box_slot_alloc (plan, sl) == 0)
                    pick = &sl;
                else
                    (void) hipGetLastError ();   /* out of memory: wait for a slot in use instead */
                break;
            }
    if (!pick)
    {
        for (BoxSlot &sl : plan->box)
            if (sl.buf && (!pick || sl.tick < pick->tick))
                pick = &sl;
        if (!pick)
            return -(int) hipErrorOutOfMemory;
        for (BoxUse &u : pick->use)
            if (u.pending && !same_queue (u.stream, stream))
            {
                HIP_TRY (hipStreamWaitEvent (stream, u.done, 0));
                u.pending = false;      /* everything later on `stream' is behind it now */
            }
    }
    int rc = box_slot_alloc (plan, *pick);
    if (rc)
        return rc;
    pick->src = src;

    /* the entry this call's event goes to: the stream's own, else an idle one, else the first -- whose call the
     * stream then waits for, so that the event recorded behind this call stands for both */
    BoxUse *use = nullptr;
    for (BoxUse &u : pick->use)
        if (u.pending && same_queue (u.stream, stream))
            use = &u;
    if (!use)
        for (BoxUse &u : pick->use)
            if (!u.pending)
            {
                use = &u;
                break;
            }
    if (!use)
    {
        use = &pick->use [0];
        HIP_TRY (hipStreamWaitEvent (stream, use->done, 0));
    }
    *slot_out = pick;
    *use_out = use;
    return 0;
}
} /* namespace */

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
    device_plan_added (device);
    *plan_out = p;
    return 0;
}

void
smol_hip_plan_destroy (SmolHipPlan *plan)
{
    if (!plan)
        return;
    const int device = plan->device;
    row_cache_release (plan);
    if (hipSetDevice (device) == hipSuccess)
        for (BoxSlot &sl : plan->box)
        {
            for (BoxUse &u : sl.use)
                if (u.done)
                {
                    if (u.pending)
                        (void) hipEventSynchronize (u.done);
                    (void) hipEventDestroy (u.done);
                }
            if (sl.buf)
                (void) hipFree (sl.buf);
        }
    delete plan;

    /* the last plan on a device gives the big staging slabs back (everything goes in smol_hip_shutdown()) */
    device_plan_removed (device);
}

int
smol_hip_plan_prepare (SmolHipPlan *plan)
{
    if (!plan)
        return SMOL_HIP_E_INVALID;
    if (!plan_may_need_scratch (plan))
        return 0;
    HIP_TRY (hipSetDevice (plan->device));
    std::lock_guard<std::mutex> guard (plan->mtx);
    return box_slot_alloc (plan, plan->box [0]);
}

int
smol_hip_plan_reserve (SmolHipPlan *plan, uint32_t images_in_flight)
{
    if (!plan || images_in_flight > (uint32_t) BOX_SLOTS)
        return SMOL_HIP_E_INVALID;
    if (!plan_may_need_scratch (plan))
        return 0;
    HIP_TRY (hipSetDevice (plan->device));
    std::lock_guard<std::mutex> guard (plan->mtx);
    for (uint32_t i = 0; i < images_in_flight; i++)
    {
        const int rc = box_slot_alloc (plan, plan->box [i]);
        if (rc)
            return rc;
    }
    return 0;
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
    if (!rows_ok (first_row, n_rows, plan->job.height_out))
        return SMOL_HIP_E_INVALID;

    HIP_TRY (hipSetDevice (plan->device));

    JobDev jd;
    fill_jobdev (jd, plan->job, pixels_in_dev, rowstride_in, dest_dev, rowstride_out, first_row, n_rows);
    JobDev probe = jd;
    if (classify (probe) != KC_BOX2)
    {
        HIP_TRY (launch_single (jd, static_cast<hipStream_t> (hip_stream), nullptr));
        return 0;
    }

    /* two passes through an intermediate image that belongs to the plan: picked, launched and marked under the
     * plan's mutex so that two calls cannot choose the same slot for different images */
    hipStream_t stream = static_cast<hipStream_t> (hip_stream);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream)
        HIP_TRY (hipStreamIsCapturing (stream, &cap));
    const bool capturing = cap != hipStreamCaptureStatusNone;
    std::lock_guard<std::mutex> guard (plan->mtx);
    BoxSlot *sl = nullptr;
    BoxUse *use = nullptr;
    int rc = plan_box_acquire (plan, pixels_in_dev, stream, capturing, &sl, &use);
    if (rc)
        return rc;
    /* the kernels index the intermediate image from the job's first source row:
     * shift the base so that row r of the image always lands at r * w_out * 8 */
    const size_t scratch_ofs = box2_tmp_offset (probe);
    void *scratch = static_cast<uint8_t *> (sl->buf) + scratch_ofs;
    HIP_TRY (launch_single (jd, stream, scratch, plan_box_bytes (plan) - scratch_ofs));
    if (use)
    {
        HIP_TRY (hipEventRecord (use->done, stream));
        use->pending = true;
        use->stream = stream;
        sl->tick = ++plan->box_tick;
    }
    return 0;
}
} /* extern "C" */
