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
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <utility>
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
        const uint32_t by = pow2_block_rows (rep);
        L::pow2 (jobs_dev, rep, dim3 ((tcols + 63) / 64, (max_rows + by - 1) / by, n_jobs), by, s);
    }
    else if (k == KC_POW2X)
    {
        /* max_w_out / max_rows carry the group's largest column / row tile counts (batches) */
        L::pow2x (jobs_dev, rep, dim3 (jobs_dev ? max_w_out : pow2x_tiles_x (rep),
                                       jobs_dev ? max_rows : pow2x_tiles_y (rep), n_jobs), s);
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
        /* rep.tmp / jobs [i].tmp point at the intermediate images; max_w_out carries the group's largest
         * pass-1 grid (box2_h_grid: x | y << 16), max_rows its largest pass-2 wave count, max_lds the largest LDS size */
        L::box2 (jobs_dev, rep, n_jobs, jobs_dev ? max_w_out : box2_h_grid (rep),
                 jobs_dev ? max_rows : box2_v_tiles (rep), jobs_dev ? max_lds : box2_h_lds_bytes (rep), s);
    }
    else if (k == KC_HSPAN)
    {
        /* max_lds carries the group's largest block count; groups share pixel size and halvings */
        L::hspan (jobs_dev, rep, dim3 (max_lds ? max_lds : hspan_blocks (rep), n_jobs), hspan_lds_bytes (rep), s);
    }
    else if (k == KC_TILE || k == KC_WTILE)
    {
        /* groups share tile_rows and LDS size */
        const uint32_t tw = 64u * rep.tile_cpl / rep.tile_lpc;
        const uint32_t tiles_x = (max_w_out + tw - 1) / tw;
        const uint32_t tiles_y = (max_rows + rep.tile_rows - 1) / rep.tile_rows;
        /* k_tile: (column tiles, row tiles, jobs); k_wtile: (tiles, jobs) */
        const dim3 grid = (k == KC_TILE) ? dim3 (tiles_x, tiles_y, n_jobs) : dim3 (tiles_x * tiles_y, n_jobs);
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
 * `scratch' is caller-owned (plan, staging lane or batch), at least box2_tmp_bytes + 64
 * bytes for the job's rows, and is indexed from the job's own first source row
 * (tmp_row0). Nothing is allocated, freed or synchronised here, so the launch is
 * asynchronous and legal under stream capture. */
hipError_t
launch_single (JobDev &jd, hipStream_t s, void *scratch = nullptr)
{
    const KClass k = classify (jd);

    if (k == KC_BOX2)
    {
        if (!scratch)
            return hipErrorInvalidValue;
        jd.tmp = static_cast<uint8_t *> (scratch);
    }
    return launch_class (k, nullptr, 1, jd, jd.w_out, jd.n_rows, s);
}

/* first_row + n_rows <= height, without wrapping */
inline bool
rows_ok (uint32_t first_row, uint32_t n_rows, uint32_t height)
{
    return first_row < height && n_rows <= height - first_row;
}

/* ---- per-device staging lanes ------------------------------------------------
 * A lane = three side streams (H2D / kernels / D2H), a pool of ordering events and
 * grow-only device slabs. A host-pointer call borrows one lane for its duration. */

constexpr int N_LANES = 4;
constexpr int MAX_DEVICES = 16;
constexpr uint32_t MAX_CHUNKS = 32;
constexpr size_t CHUNK_BYTES = 8u << 20;        /* host<->device bytes per pipeline chunk, about */

/* Ownership of a lane: a binary semaphore, not a mutex -- a lane is taken in strip_issue() and given back in
 * strip_finish(), possibly with other strips of the same thread issued in between (a device listed twice, more
 * strips than lanes), and try_lock() on a std::mutex the calling thread already owns is undefined behaviour. */
class LaneGate
{
    std::mutex m;
    std::condition_variable cv;
    bool busy = false;
public:
    bool try_acquire ()
    {
        std::lock_guard<std::mutex> g (m);
        if (busy)
            return false;
        busy = true;
        return true;
    }
    void acquire ()
    {
        std::unique_lock<std::mutex> lk (m);
        cv.wait (lk, [this] { return !busy; });
        busy = true;
    }
    void release ()
    {
        {
            std::lock_guard<std::mutex> g (m);
            busy = false;
        }
        cv.notify_one ();
    }
};

struct LaneHold
{
    LaneGate &g;
    explicit LaneHold (LaneGate &gate) : g (gate) { g.acquire (); }
    ~LaneHold () { g.release (); }
};

struct Lane
{
    LaneGate gate;
    bool ready = false;
    hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
    hipEvent_t ev_in [MAX_CHUNKS] = {}, ev_k [MAX_CHUNKS] = {};
    void *src_buf = nullptr;  size_t src_cap = 0;
    void *dst_buf = nullptr;  size_t dst_cap = 0;
    void *tmp_buf = nullptr;  size_t tmp_cap = 0;      /* intermediate image of the two-pass box filter */
};

struct DeviceState
{
    Lane lanes [N_LANES];
    std::atomic<unsigned> next { 0 };
    std::mutex ref_mtx;
    int plans = 0;                                      /* live plans on this device */
};

DeviceState g_devices [MAX_DEVICES];

int
grow (void *&buf, size_t &cap, size_t need)
{
    if (need <= cap)
        return 0;
    if (buf)
        HIP_TRY (hipFree (buf));
    buf = nullptr; cap = 0;
    HIP_TRY (hipMalloc (&buf, need));
    cap = need;
    return 0;
}

int
lane_prepare (Lane &ln, size_t src_need, size_t dst_need, size_t tmp_need)
{
    if (!ln.ready)
    {
        HIP_TRY (hipStreamCreateWithFlags (&ln.s_in, hipStreamNonBlocking));
        HIP_TRY (hipStreamCreateWithFlags (&ln.s_k, hipStreamNonBlocking));
        HIP_TRY (hipStreamCreateWithFlags (&ln.s_out, hipStreamNonBlocking));
        for (uint32_t i = 0; i < MAX_CHUNKS; i++)
        {
            HIP_TRY (hipEventCreateWithFlags (&ln.ev_in [i], hipEventDisableTiming));
            HIP_TRY (hipEventCreateWithFlags (&ln.ev_k [i], hipEventDisableTiming));
        }
        ln.ready = true;
    }
    int rc = grow (ln.src_buf, ln.src_cap, src_need);
    if (!rc) rc = grow (ln.dst_buf, ln.dst_cap, dst_need);
    if (!rc && tmp_need) rc = grow (ln.tmp_buf, ln.tmp_cap, tmp_need);
    return rc;
}

/* caller holds the lane (ln.gate) or nobody can (shutdown), and the device is current */
void
lane_release (Lane &ln)
{
    if (ln.ready)
    {
        (void) hipStreamSynchronize (ln.s_in);
        (void) hipStreamSynchronize (ln.s_k);
        (void) hipStreamSynchronize (ln.s_out);
        for (uint32_t i = 0; i < MAX_CHUNKS; i++)
        {
            if (ln.ev_in [i]) (void) hipEventDestroy (ln.ev_in [i]);
            if (ln.ev_k [i]) (void) hipEventDestroy (ln.ev_k [i]);
            ln.ev_in [i] = ln.ev_k [i] = nullptr;
        }
        (void) hipStreamDestroy (ln.s_in);
        (void) hipStreamDestroy (ln.s_k);
        (void) hipStreamDestroy (ln.s_out);
        ln.s_in = ln.s_k = ln.s_out = nullptr;
        ln.ready = false;
    }
    if (ln.src_buf) (void) hipFree (ln.src_buf);
    if (ln.dst_buf) (void) hipFree (ln.dst_buf);
    if (ln.tmp_buf) (void) hipFree (ln.tmp_buf);
    ln.src_buf = ln.dst_buf = ln.tmp_buf = nullptr;
    ln.src_cap = ln.dst_cap = ln.tmp_cap = 0;
}

void
lane_drain (Lane &ln)
{
    if (!ln.ready)
        return;
    (void) hipStreamSynchronize (ln.s_in);
    (void) hipStreamSynchronize (ln.s_k);
    (void) hipStreamSynchronize (ln.s_out);
}

void
device_release_lanes (int device)
{
    if (hipSetDevice (device) != hipSuccess)
        return;
    for (Lane &ln : g_devices [device].lanes)
    {
        LaneHold hold (ln.gate);
        lane_release (ln);
    }
}

/* When the last plan on a device goes: give back the slabs that are big (a 65535 x 4096 RGB strip
 * pins 0.8 GB), keep streams, events and ordinary-sized slabs -- one-shot callers (smol_scale_simple()
 * creates and destroys a plan per call) would otherwise pay stream + event + hipMalloc set-up on
 * every image (measured: 5 ms per 1080p call against 0.4 ms). */
/* (256 MB, not 64: the source slab of ONE 4096 x 4096 RGBA frame is 64 MB + a few bytes, and a caller that makes a
 * context per frame -- smol_scale_simple() in a loop -- paid a 64 MB hipFree + hipMalloc on every frame) */
constexpr size_t LANE_KEEP_BYTES = 256u << 20;

void
device_trim_lanes (int device)
{
    if (hipSetDevice (device) != hipSuccess)
        return;
    for (Lane &ln : g_devices [device].lanes)
    {
        LaneHold hold (ln.gate);
        if (ln.src_cap > LANE_KEEP_BYTES || ln.dst_cap > LANE_KEEP_BYTES || ln.tmp_cap > LANE_KEEP_BYTES)
            lane_drain (ln);
        if (ln.src_cap > LANE_KEEP_BYTES) { (void) hipFree (ln.src_buf); ln.src_buf = nullptr; ln.src_cap = 0; }
        if (ln.dst_cap > LANE_KEEP_BYTES) { (void) hipFree (ln.dst_buf); ln.dst_buf = nullptr; ln.dst_cap = 0; }
        if (ln.tmp_cap > LANE_KEEP_BYTES) { (void) hipFree (ln.tmp_buf); ln.tmp_buf = nullptr; ln.tmp_cap = 0; }
    }
}

bool
host_ptr_is_pinned (const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes (&attr, p) != hipSuccess)
    {
        (void) hipGetLastError ();      /* unregistered host memory reports an error: not sticky */
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

} /* namespace */

/* Intermediate image of the two-pass box filter for the device-pointer entry. A plan keeps a small ring of them:
 * calls for the SAME source image (disjoint strips, any streams) share a slot -- a boundary row two strips both
 * need gets the same bytes from both -- and calls for DIFFERENT images never meet in one: a new image takes, in
 * this order, a slot that is idle or was only used by its own stream (stream order protects it), a fresh slot, or
 * the least recently used one behind hipStreamWaitEvent on every call still using it. */
constexpr int BOX_SLOTS = 4;
constexpr int BOX_USERS = 4;

struct BoxUse
{
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;          /* recorded behind the last call this stream made on the slot */
    bool pending = false;
};

struct BoxSlot
{
    void *buf = nullptr;
    const void *src = nullptr;          /* image the slot serves (address of its row 0) */
    BoxUse use [BOX_USERS];
    uint64_t tick = 0;
};

struct SmolHipPlan
{
    SmolHipJob job;
    int device;
    std::mutex mtx;
    BoxSlot box [BOX_SLOTS];
    uint64_t box_tick = 0;
};

namespace {

size_t
plan_box_bytes (const SmolHipPlan *plan)
{
    return (size_t) plan->job.height_in * plan->job.width_out * 8 + 64;
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

/* no call of another stream is still using the slot */
bool
box_slot_free_for (BoxSlot &sl, hipStream_t stream)
{
    bool ok = true;
    for (BoxUse &u : sl.use)
    {
        if (u.pending && u.stream != stream && hipEventQuery (u.done) == hipSuccess)
            u.pending = false;
        if (u.pending && u.stream != stream)
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
            if (u.pending && u.stream != stream)
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
        if (u.pending && u.stream == stream)
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

/* One strip through a lane: issue (asynchronous) and finish (wait). The lane's mutex is
 * held from a successful issue until finish. */
struct Strip
{
    SmolHipPlan *plan = nullptr;
    const uint8_t *src = nullptr;       /* host rows, or device rows on GPU `src_device' */
    int src_device = -1;                /* -1: host memory */
    uint32_t rowstride_in = 0;
    uint8_t *dest = nullptr;            /* host: output row `first' at offset 0 */
    uint32_t rowstride_out = 0;
    uint32_t first = 0, n = 0;
    Lane *lane = nullptr;               /* set while issued */
    int rc = 0;
};

int
strip_issue_locked (Strip &st, Lane &ln)
{
    SmolHipPlan *plan = st.plan;
    const SmolHipJob &job = plan->job;
    uint32_t r0, r1;
    int rc = smol_hip_input_row_range (&job, st.first, st.n, &r0, &r1);
    if (rc)
        return rc;

    const TypeInfo &ti = k_types [job.pixel_type_in];
    const TypeInfo &to = k_types [job.pixel_type_out];
    const size_t in_row_bytes = (size_t) job.width_in * ti.bpp;
    const size_t out_row_bytes = (size_t) job.width_out * to.bpp;
    const size_t n_in_rows = (size_t) r1 - r0 + 1;
    const bool peer = st.src_device >= 0;
    /* How rows lie in the device slabs. Rows that are (nearly) back to back on the host travel as
     * ONE linear copy per chunk and keep the caller's stride and 16-byte phase on the device -- a
     * linear DMA runs at the full link rate, a 2-D copy of odd-width rows at a third of it
     * (profiles/r02a: 65535-pixel RGB rows, 17 vs 50+ GB/s); the kernels take any alignment. Rows
     * with real gaps between them are copied row-wise and re-pitched to 16-byte multiples (the
     * reference copies odd rows to an aligned temporary: smolscale-generic.c:1623-1634). A slab
     * from another GPU is always one linear peer copy. */
    const bool lin_in = peer || ((size_t) st.rowstride_in >= in_row_bytes
                                 && ((size_t) st.rowstride_in - in_row_bytes) * 8 <= in_row_bytes);
    const bool lin_out = (size_t) st.rowstride_out == out_row_bytes;
    const size_t src_pitch = lin_in ? st.rowstride_in : align_up (in_row_bytes, 16);
    const size_t dst_pitch = lin_out ? out_row_bytes : align_up (out_row_bytes, 16);
    const size_t src_skew = lin_in ? ((uintptr_t) (st.src + (size_t) r0 * st.rowstride_in) & 15u) : 0;
    const size_t dst_skew = lin_out ? ((uintptr_t) st.dest & 15u) : 0;

    if (src_pitch > 0xffffffffu || dst_pitch > 0xffffffffu)
        return SMOL_HIP_E_INVALID;

    HIP_TRY (hipSetDevice (plan->device));

    /* the two-pass box filter keeps its intermediate image in the lane */
    size_t tmp_need = 0;
    {
        JobDev probe;
        fill_jobdev (probe, job, ln.src_buf ? ln.src_buf : (void *) 16, (uint32_t) src_pitch,
                     ln.dst_buf ? ln.dst_buf : (void *) 16, (uint32_t) dst_pitch, st.first, st.n);
        if (box2_setup (probe))
            tmp_need = box2_tmp_bytes (probe) + 64;
    }
    rc = lane_prepare (ln, n_in_rows * src_pitch + 64 + 16, (size_t) st.n * dst_pitch + 64 + 16, tmp_need);
    if (rc)
        return rc;

    /* chunks of output rows: about CHUNK_BYTES of host<->device traffic each */
    const size_t total = n_in_rows * in_row_bytes + (size_t) st.n * out_row_bytes;
    uint32_t n_chunks = (uint32_t) ((total + CHUNK_BYTES - 1) / CHUNK_BYTES);
    if (n_chunks > MAX_CHUNKS) n_chunks = MAX_CHUNKS;
    if (n_chunks > st.n) n_chunks = st.n;
    if (n_chunks < 1) n_chunks = 1;
    const uint32_t rows_per_chunk = (st.n + n_chunks - 1) / n_chunks;

    uint8_t *slab = static_cast<uint8_t *> (ln.src_buf) + src_skew;
    /* Row 0 of the virtual source image lies r0 rows before the slab. */
    const uint8_t *virt_src = slab - (size_t) r0 * src_pitch;
    uint32_t copied_to = r0;            /* input rows [r0, copied_to) are on their way */
    hipError_t e = hipSuccess;

    for (uint32_t c = 0, done = 0; done < st.n && e == hipSuccess; c++)
    {
        const uint32_t rows = (st.n - done < rows_per_chunk) ? st.n - done : rows_per_chunk;
        uint32_t a, b;

        rc = smol_hip_input_row_range (&job, st.first + done, rows, &a, &b);
        if (rc)
            break;
        if (b + 1 > copied_to)
        {
            const uint32_t from = copied_to, cnt = b + 1 - copied_to;
            uint8_t *d = slab + (size_t) (from - r0) * src_pitch;
            const uint8_t *s = st.src + (size_t) from * st.rowstride_in;

            if (lin_in)
            {
                /* whole rows incl. the gaps between them, except after the very last one */
                const size_t bytes = (size_t) (cnt - 1) * st.rowstride_in + in_row_bytes;
                e = peer ? hipMemcpyPeerAsync (d, plan->device, s, st.src_device, bytes, ln.s_in)
                         : hipMemcpyAsync (d, s, bytes, hipMemcpyHostToDevice, ln.s_in);
            }
            else
                e = hipMemcpy2DAsync (d, src_pitch, s, st.rowstride_in, in_row_bytes, cnt,
                                      hipMemcpyHostToDevice, ln.s_in);
            copied_to = b + 1;
            if (e != hipSuccess)
                break;
        }
        if ((e = hipEventRecord (ln.ev_in [c], ln.s_in)) != hipSuccess) break;
        if ((e = hipStreamWaitEvent (ln.s_k, ln.ev_in [c], 0)) != hipSuccess) break;

        uint8_t *dchunk = static_cast<uint8_t *> (ln.dst_buf) + dst_skew + (size_t) done * dst_pitch;
        JobDev jd;
        fill_jobdev (jd, job, virt_src, (uint32_t) src_pitch, dchunk, (uint32_t) dst_pitch, st.first + done, rows);
        if ((e = launch_single (jd, ln.s_k, ln.tmp_buf)) != hipSuccess) break;
        if ((e = hipEventRecord (ln.ev_k [c], ln.s_k)) != hipSuccess) break;
        if ((e = hipStreamWaitEvent (ln.s_out, ln.ev_k [c], 0)) != hipSuccess) break;
        if (lin_out)
            e = hipMemcpyAsync (st.dest + (size_t) done * st.rowstride_out, dchunk, (size_t) rows * out_row_bytes,
                                hipMemcpyDeviceToHost, ln.s_out);
        else
            e = hipMemcpy2DAsync (st.dest + (size_t) done * st.rowstride_out, st.rowstride_out, dchunk, dst_pitch,
                                  out_row_bytes, rows, hipMemcpyDeviceToHost, ln.s_out);
        done += rows;
    }

    if (rc || e != hipSuccess)
    {
        lane_drain (ln);                /* nothing of the caller's memory stays in flight */
        return rc ? rc : -(int) e;
    }
    return 0;
}

/* Borrow a lane of the plan's device and issue the strip. `may_block': wait for a lane
 * when all are taken (otherwise return 1 = try again later). */
int
strip_issue (Strip &st, bool may_block)
{
    DeviceState &ds = g_devices [st.plan->device];
    const unsigned start = ds.next.fetch_add (1);
    Lane *ln = nullptr;

    for (int i = 0; i < N_LANES && !ln; i++)
    {
        Lane &cand = ds.lanes [(start + i) % N_LANES];
        if (cand.gate.try_acquire ())
            ln = &cand;
    }
    if (!ln)
    {
        if (!may_block)
            return 1;
        ln = &ds.lanes [start % N_LANES];
        ln->gate.acquire ();
    }

    st.rc = strip_issue_locked (st, *ln);
    if (st.rc)
    {
        ln->gate.release ();
        return st.rc;
    }
    st.lane = ln;
    return 0;
}

int
strip_finish (Strip &st)
{
    if (!st.lane)
        return st.rc;
    Lane &ln = *st.lane;
    hipError_t e = hipSetDevice (st.plan->device);
    if (e == hipSuccess)
        e = hipStreamSynchronize (ln.s_out);    /* the last D2H follows every kernel, every kernel its H2D */
    if (e != hipSuccess)
        lane_drain (ln);
    st.lane = nullptr;
    ln.gate.release ();
    if (e != hipSuccess)
        st.rc = -(int) e;
    return st.rc;
}

/* ---- persistent worker threads (pageable host memory: copies block their caller) ---- */

struct StripCall
{
    Strip *strips;
    uint32_t n;
    std::atomic<uint32_t> next { 0 };
    std::atomic<uint32_t> done { 0 };
    std::mutex mtx;
    std::condition_variable cv;
};

void
strip_call_help (StripCall &call)
{
    for (;;)
    {
        const uint32_t i = call.next.fetch_add (1);
        if (i >= call.n)
            return;
        Strip &st = call.strips [i];
        if (strip_issue (st, true) == 0)
            strip_finish (st);
        if (call.done.fetch_add (1) + 1 == call.n)
        {
            std::lock_guard<std::mutex> g (call.mtx);
            call.cv.notify_all ();
        }
    }
}

struct WorkerPool
{
    std::mutex mtx;
    std::condition_variable cv;
    std::deque<StripCall *> queue;      /* calls that still have unclaimed strips */
    std::vector<std::thread> threads;
    bool stop = false;

    void run ()
    {
        for (;;)
        {
            StripCall *call;
            {
                std::unique_lock<std::mutex> lk (mtx);
                for (;;)
                {
                    while (!queue.empty () && queue.front ()->next.load () >= queue.front ()->n)
                        queue.pop_front ();             /* nothing left to claim in it */
                    if (stop)
                        return;
                    if (!queue.empty ())
                        break;
                    cv.wait (lk);
                }
                call = queue.front ();
                call_enter (call);      /* under `mtx': the owner cannot retire it in between */
            }
            strip_call_help (*call);
            call_leave (call);
        }
    }

    /* owner-side lifetime: a call may be destroyed once no helper is inside it */
    std::mutex life_mtx;
    std::condition_variable life_cv;
    std::vector<std::pair<StripCall *, int>> inside;

    void call_enter (StripCall *c)
    {
        std::lock_guard<std::mutex> g (life_mtx);
        for (auto &p : inside)
            if (p.first == c) { p.second++; return; }
        inside.emplace_back (c, 1);
    }
    void call_leave (StripCall *c)
    {
        std::lock_guard<std::mutex> g (life_mtx);
        for (size_t i = 0; i < inside.size (); i++)
            if (inside [i].first == c && --inside [i].second == 0)
            {
                inside.erase (inside.begin () + (long) i);
                break;
            }
        life_cv.notify_all ();
    }
    void call_retire (StripCall *c)
    {
        {
            std::lock_guard<std::mutex> g (mtx);
            for (size_t i = 0; i < queue.size (); i++)
                if (queue [i] == c) { queue.erase (queue.begin () + (long) i); break; }
        }
        std::unique_lock<std::mutex> lk (life_mtx);
        life_cv.wait (lk, [&] {
            for (auto &p : inside) if (p.first == c) return false;
            return true; });
    }
};

std::mutex g_pool_mtx;
WorkerPool *g_pool = nullptr;           /* heap object, never destroyed at exit (threads may be parked in it) */

WorkerPool *
pool_get (uint32_t want_threads)
{
    std::lock_guard<std::mutex> g (g_pool_mtx);
    if (!g_pool)
        g_pool = new (std::nothrow) WorkerPool;
    if (!g_pool)
        return nullptr;
    if (want_threads > (uint32_t) MAX_DEVICES)
        want_threads = MAX_DEVICES;
    try
    {
        while (g_pool->threads.size () < want_threads)
            g_pool->threads.emplace_back ([p = g_pool] { p->run (); });
    }
    catch (...)
    {
        /* no thread available: the caller works through its strips itself */
    }
    return g_pool;
}

void
pool_shutdown ()
{
    WorkerPool *p;
    {
        std::lock_guard<std::mutex> g (g_pool_mtx);
        p = g_pool;
        g_pool = nullptr;
    }
    if (!p)
        return;
    {
        std::lock_guard<std::mutex> g (p->mtx);
        p->stop = true;
    }
    p->cv.notify_all ();
    for (std::thread &t : p->threads)
        t.join ();
    delete p;
}

int
run_strips (Strip *strips, uint32_t n, bool pinned)
{
    int rc = 0;

    if (n == 1)
    {
        rc = strip_issue (strips [0], true);
        return rc ? rc : strip_finish (strips [0]);
    }

    if (pinned)
    {
        /* one thread drives every device: issue all strips, then wait for each. When a
         * device is listed more often than it has lanes, finish what is pending first. */
        uint32_t issued_from = 0;
        for (uint32_t i = 0; i < n; i++)
        {
            int r = strip_issue (strips [i], false);
            if (r == 1)
            {
                for (uint32_t j = issued_from; j < i; j++)
                    if (strip_finish (strips [j]) && !rc)
                        rc = strips [j].rc;
                issued_from = i;
                r = strip_issue (strips [i], true);
            }
            if (r && !rc)
                rc = r;
        }
        for (uint32_t j = issued_from; j < n; j++)
            if (strip_finish (strips [j]) && !rc)
                rc = strips [j].rc;
        return rc;
    }

    StripCall call;
    call.strips = strips;
    call.n = n;
    WorkerPool *pool = pool_get (n - 1);
    if (pool)
    {
        {
            std::lock_guard<std::mutex> g (pool->mtx);
            pool->queue.push_back (&call);
        }
        pool->cv.notify_all ();
    }
    strip_call_help (call);             /* the caller works too: progress never depends on the pool */
    {
        std::unique_lock<std::mutex> lk (call.mtx);
        call.cv.wait (lk, [&] { return call.done.load () == call.n; });
    }
    if (pool)
        pool->call_retire (&call);
    for (uint32_t i = 0; i < n; i++)
        if (strips [i].rc && !rc)
            rc = strips [i].rc;
    return rc;
}

} /* namespace */

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
    {
        std::lock_guard<std::mutex> g (g_devices [device].ref_mtx);
        g_devices [device].plans++;
    }
    *plan_out = p;
    return 0;
}

void
smol_hip_plan_destroy (SmolHipPlan *plan)
{
    if (!plan)
        return;
    const int device = plan->device;
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
    bool last;
    {
        std::lock_guard<std::mutex> g (g_devices [device].ref_mtx);
        last = --g_devices [device].plans == 0;
    }
    if (last)
        device_trim_lanes (device);
}

int
smol_hip_plan_prepare (SmolHipPlan *plan)
{
    if (!plan)
        return SMOL_HIP_E_INVALID;
    const SmolHipJob &j = plan->job;
    if (j.filter_h != SMOL_HIP_FILTER_BOX || j.filter_v != SMOL_HIP_FILTER_BOX)
        return 0;
    HIP_TRY (hipSetDevice (plan->device));
    std::lock_guard<std::mutex> guard (plan->mtx);
    return box_slot_alloc (plan, plan->box [0]);
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
    void *scratch = static_cast<uint8_t *> (sl->buf) + (size_t) probe.tmp_row0 * probe.w_out * 8;
    HIP_TRY (launch_single (jd, stream, scratch));
    if (use)
    {
        HIP_TRY (hipEventRecord (use->done, stream));
        use->pending = true;
        use->stream = stream;
        sl->tick = ++plan->box_tick;
    }
    return 0;
}

namespace {

/* field by field: padding and the reserved bytes of a caller-built job carry no meaning */
bool
same_job (const SmolHipJob &a, const SmolHipJob &b)
{
    return a.width_in == b.width_in && a.height_in == b.height_in && a.width_out == b.width_out
           && a.height_out == b.height_out && a.pixel_type_in == b.pixel_type_in && a.pixel_type_out == b.pixel_type_out
           && a.filter_h == b.filter_h && a.filter_v == b.filter_v && a.halvings_h == b.halvings_h
           && a.halvings_v == b.halvings_v && a.wide == b.wide && a.premul16 == b.premul16 && a.linear == b.linear
           && a.quirk24 == b.quirk24 && a.span_mul_x == b.span_mul_x && a.span_mul_y == b.span_mul_y
           && a.frac0_x == b.frac0_x && a.step_x == b.step_x && a.frac0_y == b.frac0_y && a.step_y == b.step_y;
}

} /* namespace */

int
smol_hip_scale_rows_host (SmolHipPlan *plan,
                          const void *pixels_in_host, uint32_t rowstride_in,
                          void *dest_host, uint32_t rowstride_out,
                          uint32_t first_row, uint32_t n_rows)
{
    return smol_hip_scale_rows_host_multi (&plan, 1, pixels_in_host, rowstride_in,
                                           dest_host, rowstride_out, first_row, n_rows);
}

int
smol_hip_scale_rows_host_multi (SmolHipPlan *const *plans, uint32_t n_plans,
                                const void *pixels_in_host, uint32_t rowstride_in,
                                void *dest_host, uint32_t rowstride_out,
                                uint32_t first_row, uint32_t n_rows)
{
    if (!plans || !n_plans || n_plans > (uint32_t) MAX_DEVICES || !pixels_in_host || !dest_host)
        return SMOL_HIP_E_INVALID;
    for (uint32_t i = 0; i < n_plans; i++)
        if (!plans [i] || !same_job (plans [i]->job, plans [0]->job))
            return SMOL_HIP_E_INVALID;  /* every strip must belong to the same job (smol_hip.h) */
    if (n_rows == 0)
        return 0;
    if (!rows_ok (first_row, n_rows, plans [0]->job.height_out))
        return SMOL_HIP_E_INVALID;

    Strip strips [MAX_DEVICES];
    const uint32_t per = (n_rows + n_plans - 1) / n_plans;
    uint32_t n_strips = 0, done = 0;

    for (uint32_t i = 0; i < n_plans && done < n_rows; i++)
    {
        Strip &st = strips [n_strips++];
        st.plan = plans [i];
        st.src = static_cast<const uint8_t *> (pixels_in_host);
        st.rowstride_in = rowstride_in;
        st.dest = static_cast<uint8_t *> (dest_host) + (size_t) rowstride_out * done;
        st.rowstride_out = rowstride_out;
        st.first = first_row + done;
        st.n = (n_rows - done < per) ? n_rows - done : per;
        done += st.n;
    }

    bool pinned = false;
    if (n_strips > 1)
    {
        HIP_TRY (hipSetDevice (plans [0]->device));
        pinned = host_ptr_is_pinned (pixels_in_host) && host_ptr_is_pinned (dest_host);
    }
    return run_strips (strips, n_strips, pinned);
}

int
smol_hip_scale_rows_peer (SmolHipPlan *plan, int src_device,
                          const void *pixels_in_dev, uint32_t rowstride_in,
                          void *dest_host, uint32_t rowstride_out,
                          uint32_t first_row, uint32_t n_rows)
{
    if (!plan || !pixels_in_dev || !dest_host || src_device < 0 || src_device >= smol_hip_device_count ())
        return SMOL_HIP_E_INVALID;
    if (n_rows == 0)
        return 0;
    if (!rows_ok (first_row, n_rows, plan->job.height_out))
        return SMOL_HIP_E_INVALID;

    if (src_device != plan->device)
    {
        int can = 0;
        HIP_TRY (hipDeviceCanAccessPeer (&can, plan->device, src_device));
        if (can)
        {
            HIP_TRY (hipSetDevice (plan->device));
            const hipError_t e = hipDeviceEnablePeerAccess (src_device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                return -(int) e;
            (void) hipGetLastError ();
        }
        /* without peer access the runtime stages the copy through the host */
    }

    Strip st;
    st.plan = plan;
    st.src = static_cast<const uint8_t *> (pixels_in_dev);
    st.src_device = src_device;
    st.rowstride_in = rowstride_in;
    st.dest = static_cast<uint8_t *> (dest_host);
    st.rowstride_out = rowstride_out;
    st.first = first_row;
    st.n = n_rows;
    return run_strips (&st, 1, false);
}

int
smol_hip_host_register (void *host_ptr, uint64_t n_bytes)
{
    if (!host_ptr || !n_bytes)
        return SMOL_HIP_E_INVALID;
    HIP_TRY (hipHostRegister (host_ptr, n_bytes, hipHostRegisterPortable));
    return 0;
}

int
smol_hip_host_unregister (void *host_ptr)
{
    if (!host_ptr)
        return SMOL_HIP_E_INVALID;
    HIP_TRY (hipHostUnregister (host_ptr));
    return 0;
}

void
smol_hip_shutdown (void)
{
    pool_shutdown ();
    const int n = smol_hip_device_count ();
    for (int d = 0; d < n && d < MAX_DEVICES; d++)
        device_release_lanes (d);
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

    std::vector<JobDev> jobs;
    std::vector<KClass> cls;
    jobs.reserve (n_images);
    cls.reserve (n_images);

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
            if (g.k == KC_DOWN && (jobs [j].halv_h != g.rep.halv_h || jobs [j].tile_cpl != g.rep.tile_cpl || jobs [j].down_mode != g.rep.down_mode))
                continue;
            if (g.k == KC_UP && (jobs [j].tile_aligned4 != g.rep.tile_aligned4 || jobs [j].up_wide != g.rep.up_wide
                                 || (jobs [j].tile_variant & 3u) != (g.rep.tile_variant & 3u)))
                continue;
            if (g.k == KC_BOX2 && (jobs [j].wide != g.rep.wide || jobs [j].box2_ygroup != g.rep.box2_ygroup))
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
