/* smol_rt.h -- what the translation units of the host runtime share (not part of the C-ABI).
 *
 *   smol_rt_launch.hip   planner job -> kernel launch (launch_class / launch_single)
 *   smol_rt_pool.hip     worker threads: run_items() and the parallel row copy
 *   smol_rt_lanes.hip    staging lanes, page-locked bounce rings, the chunk pipeline of the host-pointer
 *                        entries, host <-> device transfers, shutdown
 *   smol_rt_plan.hip     plans, the intermediate-image ring of the two-pass box filter, the device-pointer entry
 *   smol_rt_batch.hip    batches (descriptor tables, hipGraph form, several devices, timing helpers)
 *
 * Everything here is host code; no kernel is defined or named outside smol_launch.h. */

#pragma once

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <mutex>

#include "smol_hip.h"
#include "smol_jobdev.h"

#define HIP_TRY(expr)                                           \
    do {                                                        \
        hipError_t e_ = (expr);                                 \
        if (e_ != hipSuccess)                                   \
            return -(int) e_;                                   \
    } while (0)

namespace smolrt {

using smolhip::JobDev;
using smolhip::dispatch::KClass;

constexpr int MAX_DEVICES = 16;

/* ---- smol_rt_launch.hip ---- */

/* One kernel launch over `n_jobs' jobs that share class k and the per-class grouping keys of
 * smol_hip_batch_create(); max_* are maxima over the group (their meaning per class is spelled out at the
 * call sites in smol_rt_batch.hip). jobs_dev == nullptr: the single descriptor `rep' travels by value. */
hipError_t launch_class (KClass k, const JobDev *jobs_dev, uint32_t n_jobs, const JobDev &rep,
                         uint32_t max_w_out, uint32_t max_rows, hipStream_t s, uint32_t max_lds = 0);

/* One job, descriptor by value. The two-pass box filter needs an intermediate image: `scratch' is caller-owned
 * (plan, staging lane or batch), at least box2_tmp_bytes + 64 bytes for the job's rows, and is indexed from the
 * job's own first source row (tmp_row0). Nothing is allocated, freed or synchronised, so the launch is
 * asynchronous and legal under stream capture. *k_out (optional): the class that took the job. */
hipError_t launch_single (JobDev &jd, hipStream_t s, void *scratch = nullptr, size_t scratch_bytes = 0, KClass *k_out = nullptr);

/* first_row + n_rows <= height, without wrapping */
inline bool
rows_ok (uint32_t first_row, uint32_t n_rows, uint32_t height)
{
    return first_row < height && n_rows <= height - first_row;
}

/* ---- smol_rt_pool.hip ---- */

typedef void (*ItemFn) (void *ctx, uint32_t i);

enum PoolId { POOL_STRIPS = 0, POOL_COPY = 1 };

/* fn (ctx, i) for every i in [0, n). The calling thread takes part, so progress never depends on the pool;
 * up to `helpers' persistent threads of the pool join it. Returns when every item has run. */
void run_items (PoolId pool, ItemFn fn, void *ctx, uint32_t n, uint32_t helpers);
void pools_shutdown ();

/* n_rows rows of row_bytes bytes from src (src_pitch apart) to dst (dst_pitch apart), cut into slices that the
 * copy pool's threads and the caller move concurrently: ONE thread's memcpy does not feed a PCIe 5 x16 link. */
struct RowCopy
{
    uint8_t *dst;
    size_t dst_pitch;
    const uint8_t *src;
    size_t src_pitch;
    size_t row_bytes;
    uint32_t n_rows;
    bool stream;                /* non-temporal stores: dst is a ring slot the DMA engine reads next */
};
void copy_rows (const RowCopy &rc);

/* ---- smol_rt_lanes.hip ---- */

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

/* a plan came to / left `device' (the last one leaving returns the big staging slabs) */
void device_plan_added (int device);
void device_plan_removed (int device);

/* Host <-> device transfers that never hand a PAGEABLE pointer to the HIP runtime: pageable ranges travel
 * through a lane's page-locked bounce ring (CPU copies by the copy pool), page-locked ones directly.
 * Synchronous. */
int transfer_h2d (int device, void *dst_dev, const void *src_host, size_t n_bytes);
int transfer_d2h (int device, void *dst_host, const void *src_dev, size_t n_bytes);

/* both ends of [p, p + n_bytes) are page-locked memory the runtime knows (hipHostMalloc, hipHostRegister) */
bool host_range_is_pinned (const void *p, size_t n_bytes);

} /* namespace smolrt */

/* ---- smol_rt_plan.hip ---- */

/* Intermediate image of the two-pass box filter for the device-pointer entry: a plan keeps a small ring of them
 * (see plan_box_acquire). */
constexpr int SMOL_BOX_SLOTS = 4;
constexpr int SMOL_BOX_USERS = 4;

struct SmolBoxUse
{
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;          /* recorded behind the last call this stream made on the slot */
    bool pending = false;
};

struct SmolBoxSlot
{
    void *buf = nullptr;
    const void *src = nullptr;          /* image the slot serves (address of its row 0) */
    SmolBoxUse use [SMOL_BOX_USERS];
    uint64_t tick = 0;
};

struct SmolRowCache;                    /* smol_rt_lanes.hip: read-ahead for callers that ask for their rows a few at a time */

struct SmolHipPlan
{
    SmolHipJob job;
    int device;
    std::mutex mtx;
    SmolBoxSlot box [SMOL_BOX_SLOTS];
    uint64_t box_tick = 0;
    std::mutex rc_mtx;
    SmolRowCache *rc = nullptr;         /* made when a caller's second small call follows its first */
};

namespace smolrt {
void row_cache_release (SmolHipPlan *plan);     /* plan destruction */
void row_cache_shutdown ();                     /* smol_hip_shutdown (): the spare buffers */
}
