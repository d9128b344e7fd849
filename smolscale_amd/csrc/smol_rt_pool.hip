/* smol_rt_pool.hip -- the runtime's persistent worker threads.
 *
 * Two small pools of the same kind: one drives the strips of a multi-device call (a thread per strip, mostly
 * waiting on its device), one moves bytes between the caller's pageable memory and the page-locked bounce
 * rings (a thread per slice of a chunk, mostly in memcpy). A call is a Job on its owner's stack: items are
 * claimed with one atomic counter, the owner works on them too, and it leaves only when no helper is inside
 * the job any more. Threads are created on first use and parked in a condition variable in between; a thread
 * that has just finished a job polls for the next one for a moment first, because the chunks of one strip
 * follow each other every 100-200 us and a futex wake-up costs a good part of that. */

#include "smol_rt.h"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <new>
#include <thread>
#include <vector>

#include <immintrin.h>

namespace smolrt {

namespace {

struct Job
{
    ItemFn fn;
    void *ctx;
    uint32_t n;
    uint32_t max_helpers;
    std::atomic<uint32_t> next { 0 };
    int inside = 0;                     /* helpers working in the job; guarded by the pool's mutex */
};

void
job_work (Job &job)
{
    for (;;)
    {
        const uint32_t i = job.next.fetch_add (1, std::memory_order_relaxed);
        if (i >= job.n)
            return;
        job.fn (job.ctx, i);
    }
}

struct Pool
{
    std::mutex mtx;
    std::condition_variable cv;         /* work arrived / stop */
    std::condition_variable left_cv;    /* a helper left a job */
    std::deque<Job *> queue;
    std::vector<std::thread> threads;
    std::atomic<uint64_t> pushes { 0 };
    bool stop = false;
    uint32_t max_threads;

    explicit Pool (uint32_t cap) : max_threads (cap) { }

    /* with `mtx' held */
    Job *pick ()
    {
        for (Job *j : queue)
            if (j->next.load (std::memory_order_relaxed) < j->n && j->inside < (int) j->max_helpers)
                return j;
        return nullptr;
    }

    void thread_main ()
    {
        uint64_t seen = pushes.load ();

        for (;;)
        {
            Job *job = nullptr;
            {
                std::unique_lock<std::mutex> lk (mtx);
                for (;;)
                {
                    if (stop)
                        return;
                    seen = pushes.load ();
                    if ((job = pick ()))
                        break;
                    cv.wait (lk);
                }
                job->inside++;
            }
            job_work (*job);
            {
                std::lock_guard<std::mutex> g (mtx);
                job->inside--;
            }
            left_cv.notify_all ();

            /* stay hot for a moment: the next chunk of the same strip is about to arrive */
            const auto until = std::chrono::steady_clock::now () + std::chrono::microseconds (150);
            while (pushes.load (std::memory_order_relaxed) == seen && std::chrono::steady_clock::now () < until)
                _mm_pause ();
        }
    }

    void grow (uint32_t want)
    {
        if (want > max_threads)
            want = max_threads;
        if (stop)
            return;
        try
        {
            while (threads.size () < want)
                threads.emplace_back ([this] { thread_main (); });
        }
        catch (...)
        {
            /* no thread to be had: the owner works through its items itself */
        }
    }

    void run (ItemFn fn, void *ctx, uint32_t n, uint32_t helpers)
    {
        Job job;
        job.fn = fn;
        job.ctx = ctx;
        job.n = n;
        job.max_helpers = helpers;

        if (helpers && n > 1)
        {
            {
                std::lock_guard<std::mutex> g (mtx);
                grow (helpers);
                queue.push_back (&job);
                pushes.fetch_add (1);
            }
            cv.notify_all ();
        }
        job_work (job);
        if (helpers && n > 1)
        {
            std::unique_lock<std::mutex> lk (mtx);
            for (size_t i = 0; i < queue.size (); i++)
                if (queue [i] == &job)
                {
                    queue.erase (queue.begin () + (long) i);
                    break;
                }
            /* nobody can enter any more; whoever is inside finishes the items it claimed */
            left_cv.wait (lk, [&] { return job.inside == 0; });
        }
    }

    /* The thread vector only changes under `mtx': taken out here and joined outside the lock (a helper that is
     * leaving a job needs the lock to say so). A run() that arrives meanwhile finds `stop' set and works through its
     * items alone -- grow() makes no thread while the pool is stopping. */
    void shutdown ()
    {
        std::vector<std::thread> gone;
        {
            std::lock_guard<std::mutex> g (mtx);
            stop = true;
            gone.swap (threads);
        }
        cv.notify_all ();
        for (std::thread &t : gone)
            t.join ();
        {
            std::lock_guard<std::mutex> g (mtx);
            stop = false;
        }
    }
};

uint32_t
copy_threads_default ()
{
    /* SMOL_HIP_COPY_THREADS: helpers for the bounce copies (0 = the calling thread alone). Eight move about
     * what one PCIe 5 x16 link carries; more only helps when several devices are fed at once. */
    const char *e = getenv ("SMOL_HIP_COPY_THREADS");
    if (e && *e)
    {
        const long v = strtol (e, nullptr, 10);
        return v < 0 ? 0 : v > 64 ? 64 : (uint32_t) v;
    }
    const unsigned hw = std::thread::hardware_concurrency ();
    if (hw <= 2)
        return 1;
    return hw >= 32 ? 8 : hw / 4 + 1;
}

/* heap objects that are never destroyed at exit: threads may be parked in them */
Pool *g_pools [2];
std::mutex g_pools_mtx;
std::atomic<uint32_t> g_copy_helpers { 0 };  /* written once under g_pools_mtx, read anywhere */

Pool *
pool_get (PoolId id)
{
    std::lock_guard<std::mutex> g (g_pools_mtx);
    if (!g_pools [id])
    {
        if (id == POOL_COPY)
            g_copy_helpers = copy_threads_default ();
        const uint32_t ch = g_copy_helpers.load ();
        g_pools [id] = new (std::nothrow) Pool (id == POOL_COPY ? (ch ? ch : 1) : MAX_DEVICES);
    }
    return g_pools [id];
}

/* ---- parallel row copy ---- */

constexpr size_t COPY_SLICE = 512u << 10;       /* bytes per slice, about */
constexpr size_t COPY_INLINE = 256u << 10;      /* below this the caller copies alone */

struct CopyCall
{
    RowCopy rc;
    size_t total, slice;
};

/* Filling a ring slot the DMA engine is about to read: streaming stores. A slot written with ordinary stores
 * sits, dirty, in the caches of the cores that wrote it, and the device's reads then have to be served out of
 * those caches line by line (measured: the same strip 1.7-2.0 ms against 1.4 from cold page-locked memory);
 * non-temporal stores put the bytes where the DMA engine looks for them. */
__attribute__ ((target ("avx2"))) void
copy_stream_avx2 (uint8_t *dst, const uint8_t *src, size_t n)
{
    const size_t head = (32 - ((uintptr_t) dst & 31)) & 31;

    if (n < 256 || head > n)
    {
        memcpy (dst, src, n);
        return;
    }
    memcpy (dst, src, head);
    dst += head; src += head; n -= head;
    for (; n >= 128; n -= 128, dst += 128, src += 128)
    {
        const __m256i a = _mm256_loadu_si256 ((const __m256i *) src);
        const __m256i b = _mm256_loadu_si256 ((const __m256i *) (src + 32));
        const __m256i c = _mm256_loadu_si256 ((const __m256i *) (src + 64));
        const __m256i d = _mm256_loadu_si256 ((const __m256i *) (src + 96));
        _mm256_stream_si256 ((__m256i *) dst, a);
        _mm256_stream_si256 ((__m256i *) (dst + 32), b);
        _mm256_stream_si256 ((__m256i *) (dst + 64), c);
        _mm256_stream_si256 ((__m256i *) (dst + 96), d);
    }
    memcpy (dst, src, n);
}

const bool g_have_avx2 = __builtin_cpu_supports ("avx2");

inline void
copy_bytes (uint8_t *dst, const uint8_t *src, size_t n, bool stream)
{
    if (stream && g_have_avx2)
        copy_stream_avx2 (dst, src, n);
    else
        memcpy (dst, src, n);
}

/* bytes [a, b) of the rows laid end to end */
void
copy_range (const RowCopy &rc, size_t a, size_t b)
{
    if (rc.dst_pitch == rc.row_bytes && rc.src_pitch == rc.row_bytes)
        copy_bytes (rc.dst + a, rc.src + a, b - a, rc.stream);
    else
    {
        size_t row = a / rc.row_bytes, off = a % rc.row_bytes;
        while (a < b)
        {
            size_t n = rc.row_bytes - off;
            if (n > b - a)
                n = b - a;
            copy_bytes (rc.dst + row * rc.dst_pitch + off, rc.src + row * rc.src_pitch + off, n, rc.stream);
            a += n;
            row++;
            off = 0;
        }
    }
    if (rc.stream && g_have_avx2)
        _mm_sfence ();
}

void
copy_item (void *ctx, uint32_t i)
{
    const CopyCall &c = *static_cast<const CopyCall *> (ctx);
    const size_t a = (size_t) i * c.slice;
    const size_t b = a + c.slice < c.total ? a + c.slice : c.total;
    copy_range (c.rc, a, b);
}

} /* namespace */

void
run_items (PoolId id, ItemFn fn, void *ctx, uint32_t n, uint32_t helpers)
{
    Pool *p = helpers && n > 1 ? pool_get (id) : nullptr;

    if (p)
        p->run (fn, ctx, n, helpers);
    else
        for (uint32_t i = 0; i < n; i++)
            fn (ctx, i);
}

void
pools_shutdown ()
{
    for (int id = 0; id < 2; id++)
    {
        Pool *p;
        {
            std::lock_guard<std::mutex> g (g_pools_mtx);
            p = g_pools [id];
        }
        if (p)
            p->shutdown ();             /* the object stays: a later call re-creates its threads */
    }
}

void
copy_rows (const RowCopy &rc)
{
    const size_t total = rc.row_bytes * rc.n_rows;

    if (!total)
        return;
    if (total <= COPY_INLINE)
    {
        copy_range (rc, 0, total);
        return;
    }
    pool_get (POOL_COPY);                /* fixes g_copy_helpers */
    CopyCall c;
    c.rc = rc;
    c.total = total;
    c.slice = COPY_SLICE;
    const uint32_t n = (uint32_t) ((total + c.slice - 1) / c.slice);
    run_items (POOL_COPY, copy_item, &c, n, g_copy_helpers.load ());
}

} /* namespace smolrt */
