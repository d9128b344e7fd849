/* smol_rt_lanes.hip -- the road between host memory and the device: staging lanes, page-locked bounce rings,
 * the chunk pipeline behind the host-pointer entries (what smol_scale_batch() uses), host <-> device transfers.
 *
 * The reference reads and writes the caller's buffers with plain loads and stores (smolscale-generic.c:1613-1642,
 * smolscale.c:507-546): ANY host memory works. A GPU backend that wants the same contract must not let the HIP
 * runtime see a pageable pointer: for a large pageable range the runtime pins the caller's pages in place and
 * the GPU then accesses them, and on this pool that access faults now and then when the pages are glibc's brk
 * heap under transparent huge pages (profiles/r03k_pageable_heap_fault.txt: the process dies). So pageable
 * memory is only ever touched by the CPU here: rows are copied into a lane-owned ring of hipHostMalloc'ed
 * buffers by the copy pool's threads and DMA'd from there; results are DMA'd into a second ring and copied out.
 * Page-locked callers (hipHostMalloc, smol_hip_host_alloc, smol_hip_host_register -- recognised with
 * hipPointerGetAttributes) keep the zero-copy road: their pointers go to hipMemcpyAsync directly. */

#include "smol_rt.h"
#include "smol_rt_chunks.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <thread>
#include <vector>

using namespace smolhip::dispatch;

namespace smolrt {

namespace {

constexpr int N_LANES = 4;
constexpr uint32_t MAX_CHUNKS = 32;
constexpr size_t CHUNK_BYTES = 8u << 20;        /* host<->device bytes per pipeline chunk, about */
constexpr int BOUNCE_SLOTS = 3;
constexpr size_t BOUNCE_SLACK = 256;            /* the kernels' aligned loads may overshoot a row's end */
constexpr size_t BOUNCE_MIN = 64u << 10;

/* Runtime knobs of the host road, read once (they are not on any kernel's dispatch path):
 *   SMOL_HIP_BOUNCE=0        hand pageable pointers to the runtime as rounds 1-3 did (A/B and the fault reproducer only)
 *   SMOL_HIP_BOUNCE_KB       bytes per ring slot (default 8192 KiB)
 *   SMOL_HIP_ZEROCOPY_KB     calls that move at most this much let the kernel read and write the ring directly
 *                            over PCIe -- no DMA, one launch, one wait (default 256 KiB; 0 = off)
 *   SMOL_HIP_ZEROCOPY_PAGEABLE_KB  the same bound for calls whose memory is pageable on both sides and has to pass
 *                            through the ring anyway (default 8192 KiB, at most one ring slot)
 *   SMOL_HIP_MAPPED=0        larger calls take the DMA road of rounds 1-4 instead of the mapped road (strip_issue_locked)
 *   SMOL_HIP_MAPPED_CHUNK_KB mapped road, pageable memory: bytes a chunk moves, about (default 4096 KiB, at most one ring slot)
 *   SMOL_HIP_LINK_LANES, SMOL_HIP_STREAM_STORES, SMOL_HIP_COPY_THREADS (smol_rt_pool.hip): see where they are used */
struct Knobs
{
    bool trace = false;                 /* SMOL_HIP_TRACE=1: one line per strip on stderr saying where its time went */
    bool bounce = true;
    size_t slot_bytes = 8u << 20;
    size_t zero_copy_bytes = 256u << 10;
    size_t zero_copy_pageable_bytes = 8u << 20;
    bool mapped = true;                 /* SMOL_HIP_MAPPED=0: the DMA road of rounds 1-4 for calls above the zero-copy bounds */
    size_t mapped_chunk_bytes = 4u << 20;
    bool mapped_chunk_set = false;
    bool stream_stores = true;          /* SMOL_HIP_STREAM_STORES=0: fill the upload ring with ordinary stores */
    int link_lanes = 1;                 /* SMOL_HIP_LINK_LANES: lanes (of N_LANES) that strips of the DMA road may use */

    Knobs ()
    {
        const char *e;
        if ((e = getenv ("SMOL_HIP_TRACE")) && *e == '1')
            trace = true;
        if ((e = getenv ("SMOL_HIP_BOUNCE")) && *e == '0')
            bounce = false;
        if ((e = getenv ("SMOL_HIP_BOUNCE_KB")) && *e)
        {
            const long v = strtol (e, nullptr, 10);
            if (v >= 64 && v <= (1 << 20))
                slot_bytes = (size_t) v << 10;
        }
        if ((e = getenv ("SMOL_HIP_STREAM_STORES")) && *e == '0')
            stream_stores = false;
        if ((e = getenv ("SMOL_HIP_LINK_LANES")) && *e)
        {
            const long v = strtol (e, nullptr, 10);
            if (v >= 1 && v <= N_LANES)
                link_lanes = (int) v;
        }
        if ((e = getenv ("SMOL_HIP_ZEROCOPY_PAGEABLE_KB")) && *e)
        {
            const long v = strtol (e, nullptr, 10);
            if (v >= 0 && v <= (1 << 20))
                zero_copy_pageable_bytes = (size_t) v << 10;
        }
        if ((e = getenv ("SMOL_HIP_MAPPED")) && *e == '0')
            mapped = false;
        if ((e = getenv ("SMOL_HIP_MAPPED_CHUNK_KB")) && *e)
        {
            const long v = strtol (e, nullptr, 10);
            if (v >= 64 && v <= (1 << 20))
            {
                mapped_chunk_bytes = (size_t) v << 10;
                mapped_chunk_set = true;
            }
        }
        if ((e = getenv ("SMOL_HIP_ZEROCOPY_KB")) && *e)
        {
            const long v = strtol (e, nullptr, 10);
            if (v >= 0 && v <= (1 << 20))
                zero_copy_bytes = (size_t) v << 10;
        }
    }
};

const Knobs &
knobs ()
{
    static const Knobs k;
    return k;
}

/* where a strip's wall time went (SMOL_HIP_TRACE=1), microseconds */
struct StripTimes
{
    double copy_in = 0, copy_out = 0, wait = 0, submit = 0, total = 0;
    uint32_t pieces_in = 0, pieces_out = 0;
    std::chrono::steady_clock::time_point t0;
};

inline std::chrono::steady_clock::time_point tick () { return std::chrono::steady_clock::now (); }
inline double
since (std::chrono::steady_clock::time_point t)
{
    return std::chrono::duration<double, std::micro> (std::chrono::steady_clock::now () - t).count ();
}

struct BounceSlot
{
    hipEvent_t ev = nullptr;            /* recorded behind the slot's DMA */
    bool busy = false;
    RowCopy out = {};                   /* download ring: where the bytes go once they have arrived */
};

struct BounceRing
{
    uint8_t *base = nullptr;            /* hipHostMalloc: BOUNCE_SLOTS x (slot_bytes + BOUNCE_SLACK) */
    size_t slot_bytes = 0;
    BounceSlot slot [BOUNCE_SLOTS];
    uint32_t next = 0;                  /* the slot the next piece takes = the oldest one */

    uint8_t *buf (uint32_t i) const { return base + (size_t) i * (slot_bytes + BOUNCE_SLACK); }
};

struct Lane
{
    bool busy = false;                  /* borrowed; guarded by the device's lanes_mtx */
    int device = 0;
    bool ready = false;
    hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
    hipEvent_t ev_in [MAX_CHUNKS] = {}, ev_k [MAX_CHUNKS] = {};
    void *src_buf = nullptr;  size_t src_cap = 0;
    void *dst_buf = nullptr;  size_t dst_cap = 0;
    void *tmp_buf = nullptr;  size_t tmp_cap = 0;      /* intermediate image of the two-pass box filter */
    BounceRing in, out;
    hipStream_t s_last = nullptr;       /* the stream the strip's last operation went to (strip_finish waits for it) */
    StripTimes tm;
};

struct DeviceState
{
    Lane lanes [N_LANES];
    std::mutex lanes_mtx;                               /* who holds which lane: one waiting room per device, so that */
    std::condition_variable lanes_cv;                   /* a waiter takes WHICHEVER lane of its range comes free first */
    std::atomic<unsigned> next { 0 };
    std::mutex ref_mtx;
    int plans = 0;                                      /* live plans on this device */
};

DeviceState g_devices [MAX_DEVICES];

/* Ranges this library page-locked itself (smol_hip_host_alloc / smol_hip_host_register): known end to end without
 * asking the runtime. */
std::mutex g_pinned_mtx;
std::map<uintptr_t, size_t> g_pinned;           /* base -> bytes */

void
pinned_note (const void *p, size_t n)
{
    std::lock_guard<std::mutex> g (g_pinned_mtx);
    g_pinned [(uintptr_t) p] = n;
}

void
pinned_forget (const void *p)
{
    std::lock_guard<std::mutex> g (g_pinned_mtx);
    g_pinned.erase ((uintptr_t) p);
}

/* end of the noted range that holds address q, 0 if none does */
uintptr_t
pinned_range_end (uintptr_t q)
{
    std::lock_guard<std::mutex> g (g_pinned_mtx);
    auto it = g_pinned.upper_bound (q);
    if (it == g_pinned.begin ())
        return 0;
    --it;
    return q < it->first + it->second ? it->first + it->second : 0;
}


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
    int rc = src_need ? grow (ln.src_buf, ln.src_cap, src_need) : 0;
    if (!rc && dst_need) rc = grow (ln.dst_buf, ln.dst_cap, dst_need);
    if (!rc && tmp_need) rc = grow (ln.tmp_buf, ln.tmp_cap, tmp_need);
    return rc;
}

/* Makes the ring's slots hold at least `need' bytes each (never more than the knob, never less than one row
 * of `pitch' bytes). The lane is held and every slot is idle. */
int
ring_prepare (BounceRing &r, size_t need, size_t pitch)
{
    size_t want = need < knobs ().slot_bytes ? need : knobs ().slot_bytes;

    if (want < pitch)
        want = pitch;
    if (want <= r.slot_bytes)
        return 0;
    size_t sz = BOUNCE_MIN;
    while (sz < want)
        sz *= 2;
    if (sz > knobs ().slot_bytes && want <= knobs ().slot_bytes)
        sz = knobs ().slot_bytes;

    if (r.base)
        HIP_TRY (hipHostFree (r.base));
    r.base = nullptr;
    r.slot_bytes = 0;
    void *p = nullptr;
    HIP_TRY (hipHostMalloc (&p, BOUNCE_SLOTS * (sz + BOUNCE_SLACK), hipHostMallocDefault));
    r.base = static_cast<uint8_t *> (p);
    r.slot_bytes = sz;
    r.next = 0;
    for (BounceSlot &s : r.slot)
    {
        s.busy = false;
        if (!s.ev)
            HIP_TRY (hipEventCreateWithFlags (&s.ev, hipEventDisableTiming));
    }
    return 0;
}

void
ring_release (BounceRing &r)
{
    for (BounceSlot &s : r.slot)
    {
        if (s.ev)
            (void) hipEventDestroy (s.ev);
        s.ev = nullptr;
        s.busy = false;
    }
    if (r.base)
        (void) hipHostFree (r.base);
    r.base = nullptr;
    r.slot_bytes = 0;
    r.next = 0;
}

void
ring_forget (BounceRing &r)
{
    for (BounceSlot &s : r.slot)
        s.busy = false;
}

void
lane_drain (Lane &ln)
{
    if (!ln.ready)
        return;
    (void) hipStreamSynchronize (ln.s_in);
    (void) hipStreamSynchronize (ln.s_k);
    (void) hipStreamSynchronize (ln.s_out);
    ring_forget (ln.in);
    ring_forget (ln.out);
}

/* caller holds the lane, and the device is current */
void
lane_release (Lane &ln)
{
    if (ln.ready)
    {
        lane_drain (ln);
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
    ring_release (ln.in);
    ring_release (ln.out);
    if (ln.src_buf) (void) hipFree (ln.src_buf);
    if (ln.dst_buf) (void) hipFree (ln.dst_buf);
    if (ln.tmp_buf) (void) hipFree (ln.tmp_buf);
    ln.src_buf = ln.dst_buf = ln.tmp_buf = nullptr;
    ln.src_cap = ln.dst_cap = ln.tmp_cap = 0;
}

/* lanes [first, first + count) of a device: the first free one, in fixed order; waits for one if asked to */
Lane *
lanes_take (DeviceState &ds, int first, int count, bool may_block)
{
    std::unique_lock<std::mutex> lk (ds.lanes_mtx);
    for (;;)
    {
        for (int i = 0; i < count; i++)
        {
            Lane &ln = ds.lanes [(first + i) % N_LANES];
            if (!ln.busy)
            {
                ln.busy = true;
                ln.device = (int) (&ds - g_devices);
                return &ln;
            }
        }
        if (!may_block)
            return nullptr;
        ds.lanes_cv.wait (lk);
    }
}

void
lane_give_back (Lane &ln)
{
    DeviceState &ds = g_devices [ln.device];
    {
        std::lock_guard<std::mutex> g (ds.lanes_mtx);
        ln.busy = false;
    }
    ds.lanes_cv.notify_all ();
}

void
device_release_lanes (int device)
{
    if (hipSetDevice (device) != hipSuccess)
        return;
    DeviceState &ds = g_devices [device];
    for (int i = 0; i < N_LANES; i++)
    {
        Lane *ln = lanes_take (ds, i, 1, true);
        lane_release (*ln);
        lane_give_back (*ln);
    }
}

/* When the last plan on a device goes: give back device slabs, largest first, until the lanes of the device
 * together keep at most LANES_KEEP_BYTES (a 65535 x 4096 RGB strip pins 0.8 GB); streams, events, the bounce
 * rings and ordinary-sized slabs stay -- one-shot callers (smol_scale_simple() creates and destroys a plan per
 * call) would otherwise pay stream + event + hipMalloc set-up on every image (measured: 5 ms per 1080p call
 * against 0.4 ms; a 64 MB hipFree + hipMalloc per 4096 x 4096 frame: 2.46 against 1.40 ms). Everything goes in
 * smol_hip_shutdown(). */
constexpr size_t LANES_KEEP_BYTES = 512u << 20;

void
device_trim_lanes (int device)
{
    if (hipSetDevice (device) != hipSuccess)
        return;
    DeviceState &ds = g_devices [device];

    /* lanes that are in use right now belong to a call of a plan that is being created as we speak: left alone
     * (waiting for them here would put every caller that destroys its context behind somebody else's strip) */
    for (int round = 0; round < 3 * N_LANES; round++)
    {
        size_t total = 0, big = 0;
        int big_lane = -1, big_which = 0;

        for (int i = 0; i < N_LANES; i++)
        {
            Lane *ln = lanes_take (ds, i, 1, false);
            if (!ln)
                continue;
            const size_t caps [3] = { ln->src_cap, ln->dst_cap, ln->tmp_cap };
            for (int w = 0; w < 3; w++)
            {
                total += caps [w];
                if (caps [w] > big)
                {
                    big = caps [w];
                    big_lane = i;
                    big_which = w;
                }
            }
            lane_give_back (*ln);
        }
        Lane *ln = total > LANES_KEEP_BYTES && big_lane >= 0 ? lanes_take (ds, big_lane, 1, false) : nullptr;
        if (!ln)
            return;
        void *&buf = big_which == 0 ? ln->src_buf : big_which == 1 ? ln->dst_buf : ln->tmp_buf;
        size_t &cap = big_which == 0 ? ln->src_cap : big_which == 1 ? ln->dst_cap : ln->tmp_cap;
        if (cap == big)
        {
            lane_drain (*ln);
            if (buf)
                (void) hipFree (buf);
            buf = nullptr;
            cap = 0;
        }
        lane_give_back (*ln);
    }
}

/* The first free lane in FIXED order: a caller that makes one call after the other keeps meeting the lane whose
 * slabs and rings already have the right size (in rotation, each of its first four calls paid a hipMalloc of the
 * slabs and a hipHostMalloc of the rings: 5.9 instead of 1.8 ms per 4096 x 4096 frame). Strips of the DMA road
 * (`big') only use the first link_lanes lanes -- ONE by default: a single strip fills the PCIe link, and strips
 * side by side moved LESS than one alone (4 callers of 4096 x 4096 frames from page-locked memory: 2.95 ms per
 * frame on four lanes, 1.50-1.61 on two, 1.40 on one; pageable: 6.3 / 1.8-2.5 / 1.6-1.8;
 * profiles/r04_host_path.txt). Calls of the zero-copy road are latency, not bandwidth: they take the other lanes
 * first and any lane after that. */
Lane *
lane_borrow (int device, bool may_block, bool big)
{
    DeviceState &ds = g_devices [device];
    const int n_big = knobs ().link_lanes;

    return big ? lanes_take (ds, 0, n_big, may_block) : lanes_take (ds, n_big, N_LANES, may_block);
}

/* ---- pieces through the bounce rings ------------------------------------------- */

/* n_rows rows of row_bytes bytes, src_pitch apart in PAGEABLE host memory, to device rows dev_pitch apart:
 * slot by slot -- wait for the slot's previous DMA, fill it (copy pool), DMA it on `stream'. Returns with
 * the last DMAs in flight (the slots' events tell). */
int
bounce_upload (Lane &ln, hipStream_t stream, uint8_t *dev, size_t dev_pitch,
               const uint8_t *src, size_t src_pitch, size_t row_bytes, uint32_t n_rows)
{
    BounceRing &r = ln.in;
    const size_t per64 = r.slot_bytes / dev_pitch;
    const uint32_t per = per64 < 1 ? 1 : per64 > n_rows ? n_rows : (uint32_t) per64;

    for (uint32_t at = 0; at < n_rows; at += per)
    {
        const uint32_t m = n_rows - at < per ? n_rows - at : per;
        const uint32_t idx = r.next % BOUNCE_SLOTS;
        BounceSlot &sl = r.slot [idx];

        auto t = tick ();
        if (sl.busy)
        {
            HIP_TRY (hipEventSynchronize (sl.ev));
            sl.busy = false;
        }
        ln.tm.wait += since (t); t = tick ();
        copy_rows (RowCopy { r.buf (idx), dev_pitch, src + (size_t) at * src_pitch, src_pitch, row_bytes, m, knobs ().stream_stores });
        ln.tm.copy_in += since (t); t = tick ();
        HIP_TRY (hipMemcpyAsync (dev + (size_t) at * dev_pitch, r.buf (idx), (size_t) (m - 1) * dev_pitch + row_bytes,
                                 hipMemcpyHostToDevice, stream));
        HIP_TRY (hipEventRecord (sl.ev, stream));
        ln.tm.submit += since (t);
        ln.tm.pieces_in++;
        sl.busy = true;
        r.next++;
    }
    return 0;
}

int
bounce_drain_slot (Lane &ln, BounceSlot &sl)
{
    if (!sl.busy)
        return 0;
    auto t = tick ();
    HIP_TRY (hipEventSynchronize (sl.ev));
    ln.tm.wait += since (t); t = tick ();
    copy_rows (sl.out);
    ln.tm.copy_out += since (t);
    ln.tm.pieces_out++;
    sl.busy = false;
    return 0;
}

/* the mirror image: device rows to PAGEABLE host rows. A slot's bytes are copied out when the slot is needed
 * again, when bounce_poll() finds its DMA complete, or in bounce_drain(). */
int
bounce_download (Lane &ln, hipStream_t stream, uint8_t *dst, size_t dst_pitch,
                 const uint8_t *dev, size_t dev_pitch, size_t row_bytes, uint32_t n_rows)
{
    BounceRing &r = ln.out;
    const size_t per64 = r.slot_bytes / dev_pitch;
    const uint32_t per = per64 < 1 ? 1 : per64 > n_rows ? n_rows : (uint32_t) per64;

    for (uint32_t at = 0; at < n_rows; at += per)
    {
        const uint32_t m = n_rows - at < per ? n_rows - at : per;
        const uint32_t idx = r.next % BOUNCE_SLOTS;
        BounceSlot &sl = r.slot [idx];
        const int rc = bounce_drain_slot (ln, sl);

        if (rc)
            return rc;
        auto t = tick ();
        HIP_TRY (hipMemcpyAsync (r.buf (idx), dev + (size_t) at * dev_pitch, (size_t) (m - 1) * dev_pitch + row_bytes,
                                 hipMemcpyDeviceToHost, stream));
        HIP_TRY (hipEventRecord (sl.ev, stream));
        ln.tm.submit += since (t);
        sl.out = RowCopy { dst + (size_t) at * dst_pitch, dst_pitch, r.buf (idx), dev_pitch, row_bytes, m, false };
        sl.busy = true;
        r.next++;
    }
    return 0;
}

/* copy out what has arrived, oldest first, without waiting */
int
bounce_poll (Lane &ln)
{
    BounceRing &r = ln.out;

    for (uint32_t k = 0; k < BOUNCE_SLOTS; k++)
    {
        BounceSlot &sl = r.slot [(r.next + k) % BOUNCE_SLOTS];
        if (!sl.busy)
            continue;
        const hipError_t e = hipEventQuery (sl.ev);
        if (e == hipErrorNotReady)
        {
            (void) hipGetLastError ();
            return 0;
        }
        if (e != hipSuccess)
            return -(int) e;
        auto t = tick ();
        copy_rows (sl.out);
        ln.tm.copy_out += since (t);
        ln.tm.pieces_out++;
        sl.busy = false;
    }
    return 0;
}

int
bounce_drain (Lane &ln)
{
    BounceRing &r = ln.out;

    for (uint32_t k = 0; k < BOUNCE_SLOTS; k++)
    {
        const int rc = bounce_drain_slot (ln, r.slot [(r.next + k) % BOUNCE_SLOTS]);
        if (rc)
            return rc;
    }
    return 0;
}

/* ---- one strip through a lane ---------------------------------------------------- */

/* issue (asynchronous where the memory allows it) and finish (wait). The lane is held from a successful
 * issue until finish. */
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
    bool in_place = false;              /* issued as ONE launch on page-locked memory on both sides: nothing but the lane's
                                         * kernel stream is in flight (strip_detach may hand the lane back at once) */
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

    HIP_TRY (hipSetDevice (plan->device));
    ln.tm = StripTimes ();
    ln.tm.t0 = tick ();

    /* Which road each side takes: a call that moves little takes the ring road whatever its memory is (see
     * zero_copy below); otherwise pageable memory goes through the bounce rings, page-locked memory (and a source
     * on another GPU) straight to the DMA engines. */
    const uint8_t *src_first = st.src + (size_t) r0 * st.rowstride_in;
    const bool small = knobs ().bounce && !peer
                       && n_in_rows * align_up (in_row_bytes, 16) + (size_t) st.n * align_up (out_row_bytes, 16)
                          <= (knobs ().zero_copy_bytes < knobs ().slot_bytes ? knobs ().zero_copy_bytes : knobs ().slot_bytes);   /* one ring slot holds it */
    const bool bounce_in = knobs ().bounce && !peer && !small
                           && !host_range_is_pinned (src_first, (n_in_rows - 1) * st.rowstride_in + in_row_bytes);
    const bool bounce_out = knobs ().bounce && !small
                            && !host_range_is_pinned (st.dest, (size_t) (st.n - 1) * st.rowstride_out + out_row_bytes);

    /* The mapped road (round 5): no DMA engine at all. The kernel reads its source rows from, and writes its output
     * rows to, page-locked HOST memory across PCIe -- the caller's own rows where they are page-locked (the runtime maps
     * such memory into the device's address space; the kernels take any pitch and alignment), a ring slot the copy pool
     * fills / empties where they are pageable. Reads and writes then share the link inside one kernel instead of
     * queueing as an upload, a kernel and a download: 3-27 % less time per whole-image call in every job class
     * (1080p -> 720p 0.251 -> 0.184 ms, 4096^2 -> 1024^2 1.40 -> 1.29 ms page-locked; profiles/r05_mapped_road.txt),
     * and a 64-row call costs one launch and one wait. SMOL_HIP_MAPPED=0 brings the DMA road back. */
    void *map_src = nullptr, *map_dst = nullptr;
    bool mapped = knobs ().bounce && knobs ().mapped && !peer && !small;
    /* the device's address of a page-locked range; nullptr when it has none, or when the range (it may span several
     * registrations that lie back to back on the host: host_range_is_pinned) is not ONE range on the device as well */
    auto device_view = [] (const void *host, size_t bytes) -> void *
    {
        void *first = nullptr, *last = nullptr;
        uint8_t *h = const_cast<uint8_t *> (static_cast<const uint8_t *> (host));
        if (hipHostGetDevicePointer (&first, h, 0) != hipSuccess || !first
            || hipHostGetDevicePointer (&last, h + bytes - 1, 0) != hipSuccess
            || static_cast<uint8_t *> (last) - static_cast<uint8_t *> (first) != (ptrdiff_t) (bytes - 1))
        {
            (void) hipGetLastError ();  /* (page-locked but not mapped: the DMA road) */
            return nullptr;
        }
        return first;
    };
    if (mapped && !bounce_in && !(map_src = device_view (src_first, (n_in_rows - 1) * st.rowstride_in + in_row_bytes)))
        mapped = false;
    if (mapped && !bounce_out && !(map_dst = device_view (st.dest, (size_t) (st.n - 1) * st.rowstride_out + out_row_bytes)))
        mapped = false;

    /* How rows lie in the device slabs.
     * Direct road: rows that are (nearly) back to back on the host travel as ONE linear copy per chunk and keep
     * the caller's stride and 16-byte phase on the device -- a linear DMA runs at the full link rate, a 2-D copy
     * of odd-width rows at a third of it (profiles/r02a: 65535-pixel RGB rows, 17 vs 50+ GB/s); the kernels take
     * any alignment. Rows with real gaps between them are copied row-wise and re-pitched to 16-byte multiples
     * (the reference copies odd rows to an aligned temporary: smolscale-generic.c:1623-1634). A slab from
     * another GPU is always one linear peer copy.
     * Bounce road: the CPU copy re-pitches for free, so every DMA is linear: contiguous rows stay contiguous
     * (one memcpy per slice), anything else lands on 16-byte multiples. */
    const bool lin_in = peer || ((size_t) st.rowstride_in >= in_row_bytes
                                 && ((size_t) st.rowstride_in - in_row_bytes) * 8 <= in_row_bytes);
    const bool lin_out = (size_t) st.rowstride_out == out_row_bytes;
    size_t src_pitch, dst_pitch, src_skew, dst_skew;

    if (bounce_in || small)
    {
        src_pitch = (size_t) st.rowstride_in == in_row_bytes ? in_row_bytes : align_up (in_row_bytes, 16);
        src_skew = 0;
    }
    else
    {
        src_pitch = lin_in ? st.rowstride_in : align_up (in_row_bytes, 16);
        src_skew = lin_in ? ((uintptr_t) src_first & 15u) : 0;
    }
    if (bounce_out || small)
    {
        dst_pitch = lin_out ? out_row_bytes : align_up (out_row_bytes, 16);
        dst_skew = 0;
    }
    else
    {
        dst_pitch = lin_out ? out_row_bytes : align_up (out_row_bytes, 16);
        dst_skew = lin_out ? ((uintptr_t) st.dest & 15u) : 0;
    }
    if (mapped && !bounce_in)
    {
        src_pitch = st.rowstride_in;
        src_skew = 0;
    }
    if (mapped && !bounce_out)
    {
        dst_pitch = st.rowstride_out;
        dst_skew = 0;
    }
    if (src_pitch > 0xffffffffu || dst_pitch > 0xffffffffu)
        return SMOL_HIP_E_INVALID;

    const size_t src_bytes = n_in_rows * src_pitch, dst_bytes = (size_t) st.n * dst_pitch;

    /* A call that moves little (a caller asking for its rows a few at a time: the reference's own outer loop is
     * row by row, smolscale.c:507-546) is all latency: two DMA submissions, three events and their waits cost
     * more than the bytes. The kernel then reads its source rows from, and writes its output rows to, the
     * page-locked ring itself, across PCIe: copy in, ONE launch, one wait, copy out. */
    /* ... and so does a larger call whose memory has to pass through the ring on both sides anyway, as long as
     * it fits one slot: no DMA submissions, no events (1080p -> 720p, 64 rows per call: 54 against 105 us;
     * 4096 -> 1024, 8 rows: 38 against 71 us; profiles/r04_host_path.txt) */
    /* mapped road: chunks of output rows whose ring side(s) fit a slot; page-locked on both sides: one launch */
    uint32_t map_rows = st.n;           /* output rows per chunk */
    size_t map_in_max = 0, map_out_max = 0;
    if (mapped)
    {
        /* box filters run as two kernels per chunk, or as tiles that want many rows: whole slots (4000 x 3000 -> 400 x 300:
         * 1.24 against 1.20 ms); so do strips of more than 128 MB, where the per-chunk costs add up and nothing is left of
         * the early start (65535 x 4096 RGB8 -> 8192 x 4096, 940 MB: 17.7-18.1 ms with 4 MB chunks, 16.7 with whole slots,
         * the DMA road 16.6); everything else 4 MB (4096^2 -> 1024^2: 1.41 against 1.44 ms; profiles/r05_mapped_road.txt) */
        const bool boxy = job.filter_h == SMOL_HIP_FILTER_BOX || job.filter_v == SMOL_HIP_FILTER_BOX;
        const bool large = n_in_rows * in_row_bytes + (size_t) st.n * out_row_bytes > ((size_t) 128 << 20);
        const size_t chunk = (boxy || large) && !knobs ().mapped_chunk_set ? knobs ().slot_bytes : knobs ().mapped_chunk_bytes;
        const MappedChunks mc = mapped_chunks (job, st.first, st.n, in_row_bytes, out_row_bytes, src_pitch, dst_pitch,
                                                                                      bounce_in, bounce_out, chunk, knobs ().slot_bytes);
        if (!mc.rows)
            mapped = false;             /* (one output row reads more than a slot holds: the DMA road) */
        else
        {
            map_rows = mc.rows;
            map_in_max = mc.in_max;
            map_out_max = mc.out_max;
        }
    }

    const size_t zc_pageable = knobs ().zero_copy_pageable_bytes < knobs ().slot_bytes ? knobs ().zero_copy_pageable_bytes
                                                                                         : knobs ().slot_bytes;
    const bool zero_copy = small || (!mapped && bounce_in && bounce_out && src_bytes <= zc_pageable && dst_bytes <= zc_pageable
                                     && src_bytes + dst_bytes <= zc_pageable);

    /* the two-pass box filter keeps its intermediate image in the lane */
    /* 64bpp-class box x box picks the single pass or the two passes by the bytes a launch reads (box2_setup): decided
     * HERE, once, for the whole strip, and handed to every chunk as JobDev::box_force -- a chunk of 8 MB classified on
     * its own rows would take the two passes where the strip's probe said k_tile and sized no intermediate image
     * (8000 x 6000 -> 800 x 600 failed on a fresh lane, wrote past a smaller scratch on a used one) */
    const StripClass sc = strip_classify (job, (uint32_t) src_pitch, (uint32_t) dst_pitch, st.first, st.n);
    const size_t tmp_need = sc.tmp_bytes ? sc.tmp_bytes + 64 : 0;
    rc = lane_prepare (ln, zero_copy || mapped ? 0 : src_bytes + 64 + 16, zero_copy || mapped ? 0 : dst_bytes + 64 + 16, tmp_need);
    if (!rc && (bounce_in || zero_copy))
        rc = ring_prepare (ln.in, mapped ? map_in_max : src_bytes, src_pitch);
    if (!rc && (bounce_out || zero_copy))
        rc = ring_prepare (ln.out, mapped ? map_out_max : dst_bytes, dst_pitch);
    if (rc)
        return rc;
    ln.s_last = ln.s_k;

    if (mapped)
    {
        const uint32_t S = BOUNCE_SLOTS;
        const uint32_t n_chunks = (st.n + map_rows - 1) / map_rows;
        uint32_t retired = 0;
        hipError_t e = hipSuccess;

        auto rows_of = [&] (uint32_t i) { return st.n - i * map_rows < map_rows ? st.n - i * map_rows : map_rows; };
        /* wait for chunk i's kernel; pageable destination: its rows leave the ring */
        auto retire = [&] (uint32_t i) -> hipError_t
        {
            auto t = tick ();
            /* (a strip of one chunk waits for its stream: cheaper than an event by 4 us, which a 64-row call notices) */
            const hipError_t r = n_chunks == 1 ? hipStreamSynchronize (ln.s_k) : hipEventSynchronize (ln.ev_k [i % S]);
            ln.tm.wait += since (t);
            if (r != hipSuccess || !bounce_out)
                return r;
            t = tick ();
            copy_rows (RowCopy { st.dest + (size_t) i * map_rows * st.rowstride_out, st.rowstride_out, ln.out.buf (i % S), dst_pitch,
                                 out_row_bytes, rows_of (i), false });
            ln.tm.copy_out += since (t);
            ln.tm.pieces_out++;
            return hipSuccess;
        };

        for (uint32_t i = 0; i < n_chunks && e == hipSuccess; i++)
        {
            const uint32_t first = st.first + i * map_rows, rows = rows_of (i), slot = i % S;
            uint32_t a = 0, b = 0;
            JobDev jd;

            /* slot i % S is free once chunk i - S has left it; chunks whose kernels are done already leave now */
            while (e == hipSuccess && retired < i)
            {
                if (i - retired < S)
                {
                    if (hipEventQuery (ln.ev_k [retired % S]) != hipSuccess)
                    {
                        (void) hipGetLastError ();
                        break;
                    }
                }
                e = retire (retired++);
            }
            if (e != hipSuccess)
                break;
            if (smol_hip_input_row_range (&job, first, rows, &a, &b))
            {
                lane_drain (ln);
                return SMOL_HIP_E_INVALID;
            }
            const uint8_t *virt;        /* row 0 of the virtual source image */
            if (bounce_in)
            {
                auto t = tick ();
                copy_rows (RowCopy { ln.in.buf (slot), src_pitch, st.src + (size_t) a * st.rowstride_in, st.rowstride_in, in_row_bytes,
                                     b - a + 1, knobs ().stream_stores });
                ln.tm.copy_in += since (t);
                ln.tm.pieces_in++;
                virt = ln.in.buf (slot) - (size_t) a * src_pitch;
            }
            else
                virt = static_cast<const uint8_t *> (map_src) - (size_t) r0 * src_pitch;
            uint8_t *out = bounce_out ? ln.out.buf (slot) : static_cast<uint8_t *> (map_dst) + (size_t) i * map_rows * dst_pitch;

            fill_jobdev (jd, job, virt, (uint32_t) src_pitch, out, (uint32_t) dst_pitch, first, rows);
            strip_force (jd, sc);
            auto t = tick ();
            /* (ONE stream: a stream per ring slot, so that one chunk's head overlaps the tail of the one before, was
             * measured and is 10-30 % slower -- kernels that share the link finish later each, and the copy pool waits
             * for the oldest; profiles/r05_mapped_road.txt) */
            e = launch_single (jd, ln.s_k, ln.tmp_buf, ln.tmp_cap);
            if (e == hipSuccess && (bounce_in || bounce_out) && n_chunks > 1)
                e = hipEventRecord (ln.ev_k [slot], ln.s_k);
            ln.tm.submit += since (t);
        }
        /* (page-locked on both sides: one launch in flight, strip_finish () waits for it -- strips of several devices
         * overlap; a ring side: the CPU copies are part of the strip, everything is home on return) */
        st.in_place = !bounce_in && !bounce_out;
        if (bounce_in || bounce_out)
            while (e == hipSuccess && retired < n_chunks)
                e = retire (retired++);
        if (e != hipSuccess)
        {
            lane_drain (ln);
            return -(int) e;
        }
        return 0;
    }

    if (zero_copy)
    {
        uint8_t *hin = ln.in.buf (0), *hout = ln.out.buf (0);
        JobDev jd;

        copy_rows (RowCopy { hin, src_pitch, src_first, st.rowstride_in, in_row_bytes, (uint32_t) n_in_rows, knobs ().stream_stores });
        fill_jobdev (jd, job, hin - (size_t) r0 * src_pitch, (uint32_t) src_pitch, hout, (uint32_t) dst_pitch,
                     st.first, st.n);
        strip_force (jd, sc);
        hipError_t e = launch_single (jd, ln.s_k, ln.tmp_buf, ln.tmp_cap);
        if (e == hipSuccess)
            e = hipStreamSynchronize (ln.s_k);
        if (e != hipSuccess)
        {
            lane_drain (ln);
            return -(int) e;
        }
        copy_rows (RowCopy { st.dest, st.rowstride_out, hout, dst_pitch, out_row_bytes, st.n, false });
        return 0;
    }

    /* chunks of output rows: about CHUNK_BYTES of host<->device traffic each */
    const size_t total = n_in_rows * in_row_bytes + (size_t) st.n * out_row_bytes;
    uint32_t n_chunks = (uint32_t) ((total + CHUNK_BYTES - 1) / CHUNK_BYTES);
    if (n_chunks > MAX_CHUNKS) n_chunks = MAX_CHUNKS;
    if (n_chunks > st.n) n_chunks = st.n;
    if (n_chunks < 1) n_chunks = 1;
    const uint32_t rows_per_chunk = (st.n + n_chunks - 1) / n_chunks;
    /* a strip of ONE chunk has nothing to overlap: everything goes to one stream in order, without the events
     * and cross-stream waits (six API calls less per call; a 64-row call of a 1080p image spends most of its
     * time in such calls, not in moving its 1 MB) */
    const bool one_stream = n_chunks == 1;
    const hipStream_t s_in = one_stream ? ln.s_k : ln.s_in, s_out = one_stream ? ln.s_k : ln.s_out;
    ln.s_last = s_out;

    uint8_t *slab = static_cast<uint8_t *> (ln.src_buf) + src_skew;
    /* Row 0 of the virtual source image lies r0 rows before the slab. */
    const uint8_t *virt_src = slab - (size_t) r0 * src_pitch;
    uint32_t copied_to = r0;            /* input rows [r0, copied_to) are on their way */
    hipError_t e = hipSuccess;

    /* Bounce road, upload side: the slab's rows travel as PIECES of one ring slot each, piece i through slot
     * i % BOUNCE_SLOTS: filled by the copy pool (streaming stores), then DMA'd; the DMA of piece i runs while
     * piece i + 1 is being filled. (Measured and not kept: filling the first BOUNCE_SLOTS pieces of the NEXT
     * caller's strip on a second lane while this one still owns the link -- the copy pool is the shared resource,
     * so the strip that owns the link slows down by what the other one gains: 2-4 callers 2.1-2.9 ms per
     * 4096 x 4096 frame against 1.6-1.8 with one strip at a time; profiles/r04_host_path.txt.) */
    const size_t per_in64 = bounce_in ? ln.in.slot_bytes / src_pitch : 0;
    const uint32_t per_in = per_in64 < 1 ? 1 : per_in64 > n_in_rows ? (uint32_t) n_in_rows : (uint32_t) per_in64;
    uint32_t filled = 0, submitted = 0;

    auto piece_rows = [&] (uint32_t i) { return (uint32_t) (n_in_rows - (size_t) i * per_in < per_in ? n_in_rows - (size_t) i * per_in : per_in); };
    auto fill_piece = [&] (uint32_t i) -> int
    {
        BounceSlot &sl = ln.in.slot [i % BOUNCE_SLOTS];
        auto t = tick ();
        if (sl.busy)
        {
            HIP_TRY (hipEventSynchronize (sl.ev));
            sl.busy = false;
        }
        ln.tm.wait += since (t); t = tick ();
        copy_rows (RowCopy { ln.in.buf (i % BOUNCE_SLOTS), src_pitch, src_first + (size_t) i * per_in * st.rowstride_in,
                             st.rowstride_in, in_row_bytes, piece_rows (i), knobs ().stream_stores });
        ln.tm.copy_in += since (t);
        ln.tm.pieces_in++;
        return 0;
    };
    auto submit_piece = [&] (uint32_t i) -> int
    {
        BounceSlot &sl = ln.in.slot [i % BOUNCE_SLOTS];
        auto t = tick ();
        HIP_TRY (hipMemcpyAsync (slab + (size_t) i * per_in * src_pitch, ln.in.buf (i % BOUNCE_SLOTS),
                                 (size_t) (piece_rows (i) - 1) * src_pitch + in_row_bytes, hipMemcpyHostToDevice, s_in));
        HIP_TRY (hipEventRecord (sl.ev, s_in));
        sl.busy = true;
        ln.tm.submit += since (t);
        return 0;
    };

    for (uint32_t c = 0, done = 0; done < st.n && e == hipSuccess && !rc; c++)
    {
        const uint32_t rows = (st.n - done < rows_per_chunk) ? st.n - done : rows_per_chunk;
        uint32_t a, b;

        rc = smol_hip_input_row_range (&job, st.first + done, rows, &a, &b);
        if (rc)
            break;
        if (bounce_in)
        {
            /* every piece that holds a row up to b (a piece may bring rows of the next chunk along) */
            const uint32_t need = (b - r0) / per_in;
            while (!rc && submitted <= need)
            {
                if (filled == submitted)
                    rc = fill_piece (filled++);
                if (!rc)
                    rc = submit_piece (submitted++);
            }
            if (rc)
                break;
        }
        else if (b + 1 > copied_to)
        {
            const uint32_t from = copied_to, cnt = b + 1 - copied_to;
            uint8_t *d = slab + (size_t) (from - r0) * src_pitch;
            const uint8_t *s = st.src + (size_t) from * st.rowstride_in;

            if (lin_in)
            {
                /* whole rows incl. the gaps between them, except after the very last one */
                const size_t bytes = (size_t) (cnt - 1) * st.rowstride_in + in_row_bytes;
                e = peer ? hipMemcpyPeerAsync (d, plan->device, s, st.src_device, bytes, s_in)
                         : hipMemcpyAsync (d, s, bytes, hipMemcpyHostToDevice, s_in);
            }
            else
                e = hipMemcpy2DAsync (d, src_pitch, s, st.rowstride_in, in_row_bytes, cnt,
                                      hipMemcpyHostToDevice, s_in);
            copied_to = b + 1;
            if (e != hipSuccess)
                break;
        }
        if (!one_stream)
        {
            if ((e = hipEventRecord (ln.ev_in [c], s_in)) != hipSuccess) break;
            if ((e = hipStreamWaitEvent (ln.s_k, ln.ev_in [c], 0)) != hipSuccess) break;
        }

        uint8_t *dchunk = static_cast<uint8_t *> (ln.dst_buf) + dst_skew + (size_t) done * dst_pitch;
        JobDev jd;
        fill_jobdev (jd, job, virt_src, (uint32_t) src_pitch, dchunk, (uint32_t) dst_pitch, st.first + done, rows);
        strip_force (jd, sc);
        if ((e = launch_single (jd, ln.s_k, ln.tmp_buf, ln.tmp_cap)) != hipSuccess) break;
        if (!one_stream)
        {
            if ((e = hipEventRecord (ln.ev_k [c], ln.s_k)) != hipSuccess) break;
            if ((e = hipStreamWaitEvent (s_out, ln.ev_k [c], 0)) != hipSuccess) break;
        }
        uint8_t *hdst = st.dest + (size_t) done * st.rowstride_out;
        if (bounce_out)
        {
            rc = bounce_download (ln, s_out, hdst, st.rowstride_out, dchunk, dst_pitch, out_row_bytes, rows);
            if (!rc && !one_stream)
                rc = bounce_poll (ln);
        }
        else if (lin_out)
            e = hipMemcpyAsync (hdst, dchunk, (size_t) rows * out_row_bytes, hipMemcpyDeviceToHost, s_out);
        else
            e = hipMemcpy2DAsync (hdst, st.rowstride_out, dchunk, dst_pitch,
                                  out_row_bytes, rows, hipMemcpyDeviceToHost, s_out);
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
    /* (an upper bound of what strip_issue_locked() will call small) */
    const SmolHipJob &job = st.plan->job;
    uint32_t r0 = 0, r1 = 0;
    if (smol_hip_input_row_range (&job, st.first, st.n, &r0, &r1))
        return st.rc = SMOL_HIP_E_INVALID;
    const size_t moved = ((size_t) r1 - r0 + 1) * align_up ((size_t) job.width_in * k_types [job.pixel_type_in].bpp, 16)
                         + (size_t) st.n * align_up ((size_t) job.width_out * k_types [job.pixel_type_out].bpp, 16);
    Lane *ln = lane_borrow (st.plan->device, may_block, st.src_device >= 0 || moved > knobs ().zero_copy_bytes);

    if (!ln)
        return 1;
    st.rc = strip_issue_locked (st, *ln);
    if (st.rc)
    {
        lane_give_back (*ln);
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
    int rc = 0;

    if (e == hipSuccess)
        rc = bounce_drain (ln);                 /* what still sits in the download ring */
    auto t_sync = tick ();
    if (e == hipSuccess && !rc)
        e = hipStreamSynchronize (ln.s_last);   /* the last D2H follows every kernel, every kernel its H2D */
    if (knobs ().trace)
    {
        ln.tm.wait += since (t_sync);
        ln.tm.total = since (ln.tm.t0);
        fprintf (stderr, "smol_hip strip rows %u+%u: %.0f us = copy in %.0f (%u pieces) + copy out %.0f (%u) + waits %.0f + DMA submits %.0f + rest %.0f\n",
                 st.first, st.n, ln.tm.total, ln.tm.copy_in, ln.tm.pieces_in, ln.tm.copy_out, ln.tm.pieces_out, ln.tm.wait,
                 ln.tm.submit, ln.tm.total - ln.tm.copy_in - ln.tm.copy_out - ln.tm.wait - ln.tm.submit);
    }
    if (e != hipSuccess || rc)
        lane_drain (ln);
    ring_forget (ln.in);                        /* every upload has been consumed by a kernel that is done */
    st.lane = nullptr;
    lane_give_back (ln);
    if (e != hipSuccess)
        st.rc = -(int) e;
    else if (rc)
        st.rc = rc;
    return st.rc;
}

/* An in-place strip (Strip::in_place) left in flight: `ev' is recorded behind its kernel and the lane goes back at once --
 * whoever takes the lane next queues behind the kernel on the same stream. The caller waits for `ev' before it reads the
 * strip's rows. */
int
strip_detach (Strip &st, hipEvent_t ev)
{
    if (!st.lane)
        return st.rc;
    Lane &ln = *st.lane;
    hipError_t e = hipSetDevice (st.plan->device);
    if (e == hipSuccess)
        e = hipEventRecord (ev, ln.s_last);
    if (e != hipSuccess)
        lane_drain (ln);
    st.lane = nullptr;
    lane_give_back (ln);
    if (e != hipSuccess)
        st.rc = -(int) e;
    return st.rc;
}

void
strip_item (void *ctx, uint32_t i)
{
    Strip &st = static_cast<Strip *> (ctx) [i];

    if (strip_issue (st, true) == 0)
        strip_finish (st);
}


/* ---- read-ahead for row-at-a-time callers ------------------------------------------------------------------
 *
 * The reference's own outer loop asks for ONE output row per step and keeps the two source rows it filtered last
 * (smolscale.c:507-546, smolscale-generic.c:1648-1682): a caller looping over smol_scale_batch (ctx, dest, y, 1)
 * pays for a row. Here a call is a trip to the device -- rows in, one launch, one wait, rows out: 15-19 us however
 * little it moves, 30-44 x the whole-image call for a 1080p context asked row by row. So the
 * second small call that continues where the first one ended computes a BAND of the rows that follow (about
 * RA_BAND_BYTES of traffic) into a page-locked buffer the plan keeps, and later calls inside the band are a memcpy.
 *
 * The caller may rewrite pixels_in between calls -- nothing in smolscale.h forbids it, and a row computed ahead
 * would then be stale. The band keeps the source rows it was computed FROM (they have to pass through page-locked
 * memory anyway: the staging copy is kept instead of thrown away), and a call is served from the band only if the
 * source rows ITS output rows read (smol_hip_input_row_range) still hold the same bytes: a memcmp of two or three
 * rows for a bilinear axis. Equal bytes, equal plan: the cached rows are what a fresh computation would produce.
 * SMOL_HIP_RESIDENT_SRC=1 (or smol_hip_plan_resident_source ()) is the caller's promise that the source does not
 * change while the context lives: no comparison, bands for calls of any size, larger bands.
 * SMOL_HIP_READAHEAD=0 turns the whole thing off. */
} /* namespace (anonymous) */
} /* namespace smolrt */

struct SmolRowCache
{
    uint8_t *host = nullptr;            /* hipHostMalloc: [staged source rows][band of output rows] */
    size_t host_bytes = 0;
    bool valid = false;
    const uint8_t *src = nullptr;       /* the caller's image the band was computed from */
    uint32_t rowstride_in = 0;
    uint32_t r0 = 0, r1 = 0;            /* staged source rows */
    size_t src_pitch = 0, out_pitch = 0, out_ofs = 0;
    uint32_t first = 0, n = 0;          /* output rows in the band */
    /* the pattern: where the last call on this plan ended */
    const uint8_t *last_src = nullptr;
    uint32_t last_end = 0;
    std::thread::id last_thread;        /* ... and who made it */
    int resident = -1;                  /* -1: the environment decides */
    /* under the resident-source promise: two bands of output rows, the one calls are served from and the one after it,
     * computed while the caller consumes the first (asynchronously when the source is page-locked) */
    struct Band
    {
        uint8_t *host = nullptr;        /* hipHostMalloc: the band's output rows, out_pitch apart */
        size_t host_bytes = 0;
        uint32_t first = 0, n = 0;
        bool valid = false, pending = false;
        hipEvent_t ev = nullptr;        /* recorded behind the band's kernel while `pending' */
    } band [2];
    int cur = 0;                        /* the band calls are served from */
    bool src_pinned = false;            /* the whole source image is page-locked: bands can be left in flight */
};

namespace smolrt {
namespace {

struct RaKnobs
{
    bool on = true, resident = false;
    size_t band_bytes = 2u << 20, band_bytes_resident = 8u << 20, small_bytes = 256u << 10;
    RaKnobs ()
    {
        const char *e;
        if ((e = getenv ("SMOL_HIP_READAHEAD")) && *e == '0')
            on = false;
        if ((e = getenv ("SMOL_HIP_RESIDENT_SRC")) && *e == '1')
            resident = true;
        if ((e = getenv ("SMOL_HIP_READAHEAD_KB")) && *e)
        {
            const long v = strtol (e, nullptr, 10);
            if (v >= 64 && v <= (64 << 10))
                band_bytes = band_bytes_resident = (size_t) v << 10;
        }
    }
};

const RaKnobs &ra_knobs () { static const RaKnobs k; return k; }

/* spare page-locked buffers of destroyed plans: a caller that makes a context per frame would otherwise pay a
 * hipHostMalloc + hipHostFree of megabytes per frame */
std::mutex g_ra_pool_mtx;
struct RaSpare { uint8_t *p; size_t bytes; };
std::vector<RaSpare> g_ra_pool;
constexpr size_t RA_POOL_MAX = 4;

int
ra_buffer_get (size_t need, uint8_t **p_out, size_t *bytes_out)
{
    {
        std::lock_guard<std::mutex> g (g_ra_pool_mtx);
        for (size_t i = 0; i < g_ra_pool.size (); i++)
            if (g_ra_pool [i].bytes >= need)
            {
                *p_out = g_ra_pool [i].p;
                *bytes_out = g_ra_pool [i].bytes;
                g_ra_pool.erase (g_ra_pool.begin () + (long) i);
                return 0;
            }
    }
    size_t sz = 1u << 20;
    while (sz < need)
        sz *= 2;
    void *p = nullptr;
    HIP_TRY (hipHostMalloc (&p, sz, hipHostMallocPortable));
    pinned_note (p, sz);
    *p_out = static_cast<uint8_t *> (p);
    *bytes_out = sz;
    return 0;
}

void
ra_buffer_put (uint8_t *p, size_t bytes)
{
    if (!p)
        return;
    {
        std::lock_guard<std::mutex> g (g_ra_pool_mtx);
        if (g_ra_pool.size () < RA_POOL_MAX)
        {
            g_ra_pool.push_back (RaSpare { p, bytes });
            return;
        }
    }
    pinned_forget (p);
    (void) hipHostFree (p);
}

bool
rows_equal (const uint8_t *a, size_t a_pitch, const uint8_t *b, size_t b_pitch, size_t row_bytes, uint32_t n)
{
    if (a_pitch == row_bytes && b_pitch == row_bytes)
        return memcmp (a, b, row_bytes * n) == 0;
    for (uint32_t i = 0; i < n; i++)
        if (memcmp (a + i * a_pitch, b + i * b_pitch, row_bytes))
            return false;
    return true;
}

int run_strips (Strip *strips, uint32_t n, bool pinned);

/* Under the resident-source promise (no comparison ever): bands of about `want' bytes of traffic computed straight from the
 * caller's rows into page-locked buffers of the plan, TWO of them -- while calls are served from one (a memcpy each), the next
 * is already being computed: one launch on the caller's page-locked rows in place, left in flight (strip_detach) and waited for
 * only when a call reaches it; a pageable source is copied through the rings, which needs this thread, so its bands are
 * computed when they are asked for. Caller holds plan->rc_mtx. 1: delivered; 0: not this road; < 0: error. */
int
read_ahead_resident (SmolHipPlan *plan, SmolRowCache *c, const uint8_t *src, uint32_t rowstride_in, uint8_t *dest,
                     uint32_t rowstride_out, uint32_t first, uint32_t n, size_t per_out_row, size_t out_row, size_t want, size_t want_async)
{
    const SmolHipJob &job = plan->job;
    typedef SmolRowCache::Band Band;

    auto band_wait = [&] (Band &b) -> int
    {
        if (!b.pending)
            return 0;
        b.pending = false;
        if (hipSetDevice (plan->device) != hipSuccess || hipEventSynchronize (b.ev) != hipSuccess)
        {
            (void) hipGetLastError ();
            b.valid = false;
            return SMOL_HIP_E_INVALID;
        }
        return 0;
    };
    auto covers = [&] (const Band &b) { return b.valid && first >= b.first && first + n <= b.first + b.n; };
    auto deliver = [&] (const Band &b)
    {
        const uint8_t *from = b.host + (size_t) (first - b.first) * out_row;
        if (rowstride_out == out_row)
            memcpy (dest, from, out_row * n);
        else
            for (uint32_t i = 0; i < n; i++)
                memcpy (dest + (size_t) i * rowstride_out, from + (size_t) i * out_row, out_row);
    };
    /* rows [at, at + rows) into band b; `ahead': may be left in flight */
    auto compute = [&] (Band &b, uint32_t at, uint32_t rows, bool ahead) -> int
    {
        int rc = band_wait (b);         /* (its buffer is about to be written again) */
        b.valid = false;
        if (rc)
            return rc;
        const size_t need = (size_t) rows * out_row + 256;
        if (b.host_bytes < need)
        {
            if (hipSetDevice (plan->device) != hipSuccess)
                return SMOL_HIP_E_INVALID;
            ra_buffer_put (b.host, b.host_bytes);
            b.host = nullptr; b.host_bytes = 0;
            if (ra_buffer_get (need, &b.host, &b.host_bytes))
            {
                (void) hipGetLastError ();
                return SMOL_HIP_E_INVALID;
            }
        }
        Strip st;
        st.plan = plan;
        st.src = src;
        st.rowstride_in = rowstride_in;
        st.dest = b.host;
        st.rowstride_out = (uint32_t) out_row;
        st.first = at;
        st.n = rows;
        rc = strip_issue (st, true);
        if (rc)
            return rc < 0 ? rc : SMOL_HIP_E_INVALID;
        if (ahead && st.in_place)
        {
            if (!b.ev && hipEventCreateWithFlags (&b.ev, hipEventDisableTiming) != hipSuccess)
            {
                (void) hipGetLastError ();
                b.ev = nullptr;
            }
            if (b.ev)
            {
                rc = strip_detach (st, b.ev);
                b.pending = rc == 0;
            }
            else
                rc = strip_finish (st);
        }
        else
            rc = strip_finish (st);
        if (rc)
            return rc;
        b.first = at; b.n = rows; b.valid = true;
        return 0;
    };
    auto band_rows = [&] (uint32_t at) -> uint32_t
    {
        uint64_t rows = want / per_out_row;
        if (rows < 2ull * n)
            rows = 2ull * n;
        rows -= rows % n;               /* (whole calls: a caller that keeps its step never straddles two bands) */
        if (rows > job.height_out - at)
            rows = job.height_out - at;
        return (uint32_t) rows;
    };
    /* the band after `b', unless it is there (or on its way) already */
    auto look_ahead = [&] (const Band &b, Band &other)
    {
        const uint32_t at = b.first + b.n;
        if (at >= job.height_out || ((other.valid || other.pending) && other.first == at))
            return;
        (void) compute (other, at, band_rows (at), true);       /* (a failure here costs nothing: the call that needs the rows tries again) */
    };

    if (c->src != src || c->rowstride_in != rowstride_in)
    {
        /* another image behind the same plan: what is on file belongs to the old one */
        for (Band &b : c->band)
        {
            (void) band_wait (b);
            b.valid = false;
        }
        c->src = src; c->rowstride_in = rowstride_in;
        c->src_pinned = host_range_is_pinned (src, (size_t) (job.height_in - 1) * rowstride_in
                                                   + (size_t) job.width_in * k_types [job.pixel_type_in].bpp);
    }
    /* bands that can be left in flight: a quarter of the image each, so that only the first is waited for -- but no shorter
     * than `want_async' (a band costs a launch, a handful of runtime calls and a wait: 2 MB bands under 8-row calls ran
     * 4096^2 -> 1024^2 in 3.8 ms where 8 MB bands take 1.5) and no longer than `want' */
    if (c->src_pinned)
    {
        const size_t quarter = (size_t) job.height_out * per_out_row / 4;
        want = quarter < want_async ? want_async : quarter < want ? quarter : want;
    }

    Band &cur = c->band [c->cur], &nxt = c->band [c->cur ^ 1];
    if (covers (cur))
    {
        deliver (cur);
        /* (a pageable source: the next band is computed by a call in the band's second half -- this thread has to copy it
         * through the rings, and a caller that stops early should not have paid for it) */
        if (c->src_pinned || first + n >= cur.first + cur.n / 2)
            look_ahead (cur, nxt);
        return 1;
    }
    if (cur.valid && (nxt.valid || nxt.pending) && cur.first + cur.n == nxt.first
        && first >= cur.first && first < nxt.first && first + n > nxt.first && first + n <= nxt.first + nxt.n)
    {
        /* a call across the seam of the two bands: its first rows from the one, the rest from the other */
        if (band_wait (nxt) || !nxt.valid)
            return 0;
        const uint32_t n0 = nxt.first - first;
        for (uint32_t i = 0; i < n; i++)
            memcpy (dest + (size_t) i * rowstride_out,
                    i < n0 ? cur.host + (size_t) (first + i - cur.first) * out_row : nxt.host + (size_t) (i - n0) * out_row, out_row);
        c->cur ^= 1;
        look_ahead (nxt, cur);
        return 1;
    }
    if ((nxt.valid || nxt.pending) && first >= nxt.first && first + n <= nxt.first + nxt.n)
    {
        if (band_wait (nxt) || !nxt.valid)
            return 0;
        c->cur ^= 1;
        deliver (nxt);
        look_ahead (nxt, cur);
        return 1;
    }
    /* a miss: a band from `first' on, now; then the one after it, ahead */
    const uint32_t rows = band_rows (first);
    if (rows <= n)
        return 0;
    const int rc = compute (cur, first, rows, false);
    if (rc)
        return rc;
    deliver (cur);
    look_ahead (cur, nxt);
    return 1;
}

/* 1: the rows were delivered from (or through) the read-ahead band; 0: not this road, take the ordinary one;
 * < 0: error */
int
read_ahead (SmolHipPlan *plan, const uint8_t *src, uint32_t rowstride_in, uint8_t *dest, uint32_t rowstride_out,
            uint32_t first, uint32_t n)
{
    const RaKnobs &kn = ra_knobs ();
    const SmolHipJob &job = plan->job;
    if (!kn.on || n >= job.height_out)
        return 0;

    const TypeInfo &ti = k_types [job.pixel_type_in];
    const TypeInfo &to = k_types [job.pixel_type_out];
    const size_t in_row = (size_t) job.width_in * ti.bpp, out_row = (size_t) job.width_out * to.bpp;
    uint32_t a = 0, b = 0;
    if (smol_hip_input_row_range (&job, first, n, &a, &b))
        return 0;

    /* Callers that thread themselves over ONE context with small row batches (the reference's own worker pattern, test.c:838-883)
     * must not queue up behind a band another thread is computing under this lock (staging copy, launch, wait: up to 2-8 MB):
     * a contended lock means "not this road" (ADVICE round 5). And the pattern is per calling thread: a call from another
     * thread than the last one starts a new sequence instead of continuing -- or breaking -- somebody else's. */
    std::unique_lock<std::mutex> guard (plan->rc_mtx, std::try_to_lock);
    if (!guard.owns_lock ())
        return 0;
    SmolRowCache *c = plan->rc;
    const bool resident = c && c->resident >= 0 ? c->resident != 0 : kn.resident;
    const size_t moved = ((size_t) b - a + 1) * in_row + (size_t) n * out_row;
    if (!resident && moved > kn.small_bytes)
    {
        if (c)
            c->last_src = nullptr;      /* (the pattern is broken) */
        return 0;                       /* a call of this size is worth its own trip; comparing its rows is not */
    }
    if (!c)
    {
        c = plan->rc = new (std::nothrow) SmolRowCache;
        if (!c)
            return 0;
    }

    auto deliver = [&] () -> bool
    {
        if (!c->valid || c->src != src || c->rowstride_in != rowstride_in || first < c->first || first + n > c->first + c->n)
            return false;
        if (!rows_equal (src + (size_t) a * rowstride_in, rowstride_in, c->host + (size_t) (a - c->r0) * c->src_pitch, c->src_pitch,
                         in_row, b - a + 1))
        {
            c->valid = false;           /* the caller rewrote its source: what was computed ahead is stale */
            return false;
        }
        const uint8_t *from = c->host + c->out_ofs + (size_t) (first - c->first) * c->out_pitch;
        if (rowstride_out == out_row && c->out_pitch == out_row)
            memcpy (dest, from, out_row * n);
        else
            for (uint32_t i = 0; i < n; i++)
                memcpy (dest + (size_t) i * rowstride_out, from + (size_t) i * c->out_pitch, out_row);
        return true;
    };

    const std::thread::id me = std::this_thread::get_id ();
    const bool continues = c->last_src == src && c->last_end == first && c->last_thread == me;
    c->last_src = src;
    c->last_end = first + n;
    c->last_thread = me;
    if (resident)
    {
        /* (the first call of a sequence takes the ordinary road: a caller that asks for one strip only pays for no band) */
        const bool on_file = c->src == src && c->rowstride_in == rowstride_in
                             && (c->band [0].valid || c->band [0].pending || c->band [1].valid || c->band [1].pending);
        if (!continues && !on_file)
            return 0;
        return read_ahead_resident (plan, c, src, rowstride_in, dest, rowstride_out, first, n, moved / n + 1, out_row, kn.band_bytes_resident, kn.band_bytes);
    }
    if (deliver ())
        return 1;
    if (!continues)
        return 0;                       /* a first call, or one out of order: the ordinary road */

    /* a band of the rows from `first' on */
    const size_t per_out_row = moved / n + 1;
    const size_t want = kn.band_bytes;                  /* (the promise's bands: read_ahead_resident) */
    uint64_t rows = want / per_out_row;
    if (rows < 2ull * n)
        return 0;                       /* rows so fat that a band is hardly more than the call itself */
    if (rows > job.height_out - first)
        rows = job.height_out - first;
    if (rows <= n)
        return 0;
    uint32_t r0 = 0, r1 = 0;
    if (smol_hip_input_row_range (&job, first, (uint32_t) rows, &r0, &r1))
        return 0;
    const size_t src_pitch = align_up (in_row, 16), out_pitch = out_row;
    const size_t src_bytes = ((size_t) r1 - r0 + 1) * src_pitch + 256, out_bytes = (size_t) rows * out_pitch + 256;
    if (c->host_bytes < src_bytes + out_bytes)
    {
        if (hipSetDevice (plan->device) != hipSuccess)
            return 0;
        ra_buffer_put (c->host, c->host_bytes);
        c->host = nullptr; c->host_bytes = 0; c->valid = false;
        if (ra_buffer_get (src_bytes + out_bytes, &c->host, &c->host_bytes))
        {
            (void) hipGetLastError ();
            return 0;
        }
    }
    c->valid = false;
    copy_rows (RowCopy { c->host, src_pitch, src + (size_t) r0 * rowstride_in, rowstride_in, in_row, r1 - r0 + 1, true });

    Strip st;
    st.plan = plan;
    st.src = c->host - (size_t) r0 * src_pitch;     /* (row 0 of the virtual image; only rows r0 .. r1 are ever read) */
    st.rowstride_in = (uint32_t) src_pitch;
    st.dest = c->host + src_bytes;
    st.rowstride_out = (uint32_t) out_pitch;
    st.first = first;
    st.n = (uint32_t) rows;
    const int rc = run_strips (&st, 1, true);
    if (rc)
        return rc;
    c->src = src; c->rowstride_in = rowstride_in;
    c->r0 = r0; c->r1 = r1;
    c->src_pitch = src_pitch; c->out_pitch = out_pitch; c->out_ofs = src_bytes;
    c->first = first; c->n = (uint32_t) rows;
    c->valid = true;
    /* (delivered without another comparison: the staged rows were copied from the caller's a moment ago, inside this call) */
    const uint8_t *from = c->host + c->out_ofs;
    for (uint32_t i = 0; i < n; i++)
        memcpy (dest + (size_t) i * rowstride_out, from + (size_t) i * out_pitch, out_row);
    return 1;
}

} /* namespace */

void
row_cache_release (SmolHipPlan *plan)
{
    std::lock_guard<std::mutex> guard (plan->rc_mtx);
    if (!plan->rc)
        return;
    ra_buffer_put (plan->rc->host, plan->rc->host_bytes);
    for (SmolRowCache::Band &b : plan->rc->band)
    {
        if (b.pending || b.ev)
            (void) hipSetDevice (plan->device);
        if (b.pending)
            (void) hipEventSynchronize (b.ev);         /* (its kernel writes into the buffer that goes back now) */
        if (b.ev)
            (void) hipEventDestroy (b.ev);
        ra_buffer_put (b.host, b.host_bytes);
    }
    delete plan->rc;
    plan->rc = nullptr;
}

void
row_cache_shutdown ()
{
    std::lock_guard<std::mutex> g (g_ra_pool_mtx);
    for (RaSpare &sp : g_ra_pool)
    {
        pinned_forget (sp.p);
        (void) hipHostFree (sp.p);
    }
    g_ra_pool.clear ();
}

namespace {

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

    /* pageable memory: a strip's CPU copies block the thread that drives it, so every strip gets a thread
     * (the caller takes one too: progress never depends on the pool) */
    run_items (POOL_STRIPS, strip_item, strips, n, n - 1);
    for (uint32_t i = 0; i < n; i++)
        if (strips [i].rc && !rc)
            rc = strips [i].rc;
    return rc;
}

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

/* a linear transfer through a lane's ring on the NULL stream: like hipMemcpy it is ordered behind whatever
 * the caller queued on blocking streams before, and complete on return */
int
transfer (int device, void *dev, void *host, size_t n_bytes, bool to_device)
{
    if (!n_bytes)
        return 0;
    if (device < 0 || device >= MAX_DEVICES || !dev || !host)
        return SMOL_HIP_E_INVALID;
    HIP_TRY (hipSetDevice (device));
    if (!knobs ().bounce || host_range_is_pinned (host, n_bytes))
    {
        HIP_TRY (to_device ? hipMemcpy (dev, host, n_bytes, hipMemcpyHostToDevice)
                           : hipMemcpy (host, dev, n_bytes, hipMemcpyDeviceToHost));
        return 0;
    }

    Lane *ln = lane_borrow (device, true, n_bytes > knobs ().zero_copy_bytes);
    int rc = lane_prepare (*ln, 0, 0, 0);
    BounceRing &ring = to_device ? ln->in : ln->out;
    if (!rc)
        rc = ring_prepare (ring, n_bytes, 1);
    if (!rc)
    {
        /* pieces of one slot each: `rows' of slot_bytes plus a tail */
        const size_t piece = ring.slot_bytes;
        const uint32_t full = (uint32_t) (n_bytes / piece);
        const size_t tail = n_bytes - (size_t) full * piece;
        uint8_t *d = static_cast<uint8_t *> (dev), *h = static_cast<uint8_t *> (host);

        if (to_device)
        {
            if (full)
                rc = bounce_upload (*ln, nullptr, d, piece, h, piece, piece, full);
            if (!rc && tail)
                rc = bounce_upload (*ln, nullptr, d + (size_t) full * piece, tail, h + (size_t) full * piece, tail, tail, 1);
        }
        else
        {
            if (full)
                rc = bounce_download (*ln, nullptr, h, piece, d, piece, piece, full);
            if (!rc && tail)
                rc = bounce_download (*ln, nullptr, h + (size_t) full * piece, tail, d + (size_t) full * piece, tail, tail, 1);
            if (!rc)
                rc = bounce_drain (*ln);
        }
        if (!rc && hipStreamSynchronize (nullptr) != hipSuccess)
            rc = -(int) hipGetLastError ();
        if (rc)
            (void) hipDeviceSynchronize ();
        ring_forget (ring);
    }
    lane_give_back (*ln);
    return rc;
}

} /* namespace */

/* Is EVERY byte of [p, p + n_bytes) page-locked? The range is walked allocation by allocation: the library's own
 * registrations from its table, foreign ones (hipHostMalloc / hipHostRegister by the caller) through the runtime --
 * hipPointerGetAttributes says what the address is, hipMemGetAddressRange where its allocation ends. Rounds 3-4 asked
 * about the two END bytes only: two registrations with a pageable gap between them passed, and the gap went to the
 * DMA engine. Any doubt -- a query that fails, an allocation whose extent the runtime does
 * not tell -- means "pageable": the bounce rings are always correct. */
bool
host_range_is_pinned (const void *p, size_t n_bytes)
{
    uintptr_t q = (uintptr_t) p;
    const uintptr_t end = q + (n_bytes ? n_bytes : 1);

    for (int hops = 0; q < end; hops++)
    {
        if (hops > 64)
            return false;               /* a range stitched from that many allocations: not worth the queries */
        const uintptr_t own_end = pinned_range_end (q);
        if (own_end)
        {
            q = own_end;
            continue;
        }
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes (&attr, reinterpret_cast<const void *> (q)) != hipSuccess)
        {
            (void) hipGetLastError ();  /* unregistered host memory reports an error: not sticky */
            return false;
        }
        if (attr.type != hipMemoryTypeHost || !attr.hostPointer || !attr.devicePointer)
            return false;
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange (&base, &size, attr.devicePointer) != hipSuccess || !size)
        {
            (void) hipGetLastError ();
            return false;
        }
        /* the allocation's end, seen from the host address */
        const uintptr_t dev_end = (uintptr_t) base + size;
        if (dev_end <= (uintptr_t) attr.devicePointer)
            return false;
        q += dev_end - (uintptr_t) attr.devicePointer;
    }
    return true;
}

void
device_plan_added (int device)
{
    std::lock_guard<std::mutex> g (g_devices [device].ref_mtx);
    g_devices [device].plans++;
}

void
device_plan_removed (int device)
{
    bool last;
    {
        std::lock_guard<std::mutex> g (g_devices [device].ref_mtx);
        last = --g_devices [device].plans == 0;
    }
    if (last)
        device_trim_lanes (device);
}

int
transfer_h2d (int device, void *dst_dev, const void *src_host, size_t n_bytes)
{
    return transfer (device, dst_dev, const_cast<void *> (src_host), n_bytes, true);
}

int
transfer_d2h (int device, void *dst_host, const void *src_dev, size_t n_bytes)
{
    return transfer (device, const_cast<void *> (src_dev), dst_host, n_bytes, false);
}

} /* namespace smolrt */

using namespace smolrt;

extern "C" {

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

    if (n_plans == 1)
    {
        /* a caller that asks for its rows a few at a time: served from (or starting) a band computed ahead */
        const int ra = read_ahead (plans [0], static_cast<const uint8_t *> (pixels_in_host), rowstride_in,
                                   static_cast<uint8_t *> (dest_host), rowstride_out, first_row, n_rows);
        if (ra)
            return ra < 0 ? ra : 0;
    }

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
        const SmolHipJob &job = plans [0]->job;
        const size_t in_bytes = (size_t) (job.height_in - 1) * rowstride_in + (size_t) job.width_in * k_types [job.pixel_type_in].bpp;
        const size_t out_bytes = (size_t) (n_rows - 1) * rowstride_out + (size_t) job.width_out * k_types [job.pixel_type_out].bpp;

        HIP_TRY (hipSetDevice (plans [0]->device));
        pinned = host_range_is_pinned (pixels_in_host, in_bytes) && host_range_is_pinned (dest_host, out_bytes);
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
    pinned_note (host_ptr, n_bytes);
    return 0;
}

int
smol_hip_host_unregister (void *host_ptr)
{
    if (!host_ptr)
        return SMOL_HIP_E_INVALID;
    pinned_forget (host_ptr);
    HIP_TRY (hipHostUnregister (host_ptr));
    return 0;
}

int
smol_hip_host_alloc (uint64_t n_bytes, void **host_ptr_out)
{
    if (!host_ptr_out || !n_bytes)
        return SMOL_HIP_E_INVALID;
    HIP_TRY (hipHostMalloc (host_ptr_out, n_bytes, hipHostMallocPortable));
    pinned_note (*host_ptr_out, n_bytes);
    return 0;
}

int
smol_hip_host_free (void *host_ptr)
{
    if (!host_ptr)
        return SMOL_HIP_E_INVALID;
    pinned_forget (host_ptr);
    HIP_TRY (hipHostFree (host_ptr));
    return 0;
}

int
smol_hip_plan_resident_source (SmolHipPlan *plan, int resident)
{
    if (!plan)
        return SMOL_HIP_E_INVALID;
    std::lock_guard<std::mutex> guard (plan->rc_mtx);
    if (!plan->rc)
        plan->rc = new (std::nothrow) SmolRowCache;
    if (!plan->rc)
        return SMOL_HIP_E_INVALID;
    SmolRowCache *c = plan->rc;
    const int want = resident ? 1 : 0;
    /* Whatever was computed ahead under the old setting is dropped, in flight or not: a caller that takes the promise back,
     * rewrites its source and gives the promise again for the same pointer must not be served rows of the old bytes -- under
     * the promise nothing is compared. */
    for (SmolRowCache::Band &b : c->band)
    {
        if (b.pending)
        {
            (void) hipSetDevice (plan->device);
            (void) hipEventSynchronize (b.ev);
            (void) hipGetLastError ();
            b.pending = false;
        }
        b.valid = false;
    }
    c->valid = false;
    c->src = nullptr;
    c->last_src = nullptr;              /* (the next call starts a new sequence) */
    c->resident = want;
    return 0;
}

int
smol_hip_host_range_is_pinned (const void *host_ptr, uint64_t n_bytes)
{
    return host_ptr && host_range_is_pinned (host_ptr, (size_t) n_bytes) ? 1 : 0;
}

void
smol_hip_shutdown (void)
{
    pools_shutdown ();
    row_cache_shutdown ();
    const int n = smol_hip_device_count ();
    for (int d = 0; d < n && d < MAX_DEVICES; d++)
        device_release_lanes (d);
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
    return transfer_h2d (device, dst_dev, src_host, n_bytes);
}

int
smol_hip_memcpy_d2h (int device, void *dst_host, const void *src_dev, uint64_t n_bytes)
{
    return transfer_d2h (device, dst_host, src_dev, n_bytes);
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
