/* smol_hip.h -- C-ABI boundary of the HIP (gfx950 / MI355X) resampling backend.
 *
 * Plain C types only: pointers, fixed-width integers and POD structs. This is
 * what the host-side dispatcher binds. In the reference the same seam is the
 * row-function table `SmolImplementation' (reference smolscale-private.h:270-278)
 * filled by _smol_get_generic_implementation() (smolscale-generic.c:2498) and
 * consumed by do_rows() (smolscale.c:507-546); a GPU wants whole row RANGES, so
 * the seam here is one call per smol_scale_batch() instead of one per row.
 * INTEGRATION.md shows the patch a reference maintainer would apply.
 *
 * All functions return 0 on success or a negative value (-(hipError_t), or
 * SMOL_HIP_E_*) on failure; nothing here falls back to a CPU path. */

#ifndef SMOLSCALE_AMD_SMOL_HIP_H
#define SMOLSCALE_AMD_SMOL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMOL_HIP_E_INVALID      (-10001)   /* bad argument                           */
#define SMOL_HIP_E_UNSUPPORTED  (-10002)   /* job outside what the kernels implement */
#define SMOL_HIP_E_NO_DEVICE    (-10003)   /* no usable GPU                          */

/* Filter classes chosen per axis (replaces reference enum SmolFilterType,
 * smolscale-private.h:100-116; BILINEAR carries its halvings separately and
 * only 0..2 are reachable, see smol_hip_plan_job()). */
enum
{
    SMOL_HIP_FILTER_COPY = 0,
    SMOL_HIP_FILTER_ONE = 1,
    SMOL_HIP_FILTER_BILINEAR = 2,
    SMOL_HIP_FILTER_BOX = 3
};

/* Everything the kernels need to know about one scaling job: the result of the
 * planner (replaces the reference's pick_filter_params() smolscale.c:427-478,
 * smol_scale_init() :816-870, get_implementations() :725-783 and the precalc
 * tables of smolscale-generic.c:14-179, which are kept in closed form here:
 * sample j of a bilinear axis sits at fixed-point position frac0 + j*step
 * (32.32), box k ends at (k+1)*step (16.16)). */
typedef struct
{
    uint32_t width_in, height_in;
    uint32_t width_out, height_out;
    uint8_t  pixel_type_in, pixel_type_out;   /* SmolPixelType                        */
    uint8_t  filter_h, filter_v;              /* SMOL_HIP_FILTER_*                    */
    uint8_t  halvings_h, halvings_v;          /* bilinear only, 0..2                  */
    uint8_t  wide;                            /* 0: 64bpp-class math, 1: 128bpp-class */
    uint8_t  premul16;                        /* unassociated -> unassociated         */
    uint8_t  linear;                          /* sRGB-linear internal representation  */
    uint8_t  quirk24;                         /* 24bpp sRGB output via `123' packer   */
    uint8_t  reserved [2];
    uint32_t span_mul_x, span_mul_y;          /* box normalisers                      */
    uint64_t frac0_x, step_x;                 /* bilinear: 32.32; box: step is 16.16  */
    uint64_t frac0_y, step_y;
}
SmolHipJob;

/* One image of a batched launch (device pointers). Output row `first_row' is
 * written at dst + 0; src points at input row 0. */
typedef struct
{
    SmolHipJob job;
    const void *src;
    void *dst;
    uint32_t rowstride_in, rowstride_out;     /* bytes, any alignment */
    uint32_t first_row, n_rows;
}
SmolHipImage;

typedef struct SmolHipPlan SmolHipPlan;

/* ---- host-only helpers (no GPU needed) ---------------------------------- */

/* Planner. Fills *job for scaling (type_in, w_in x h_in) -> (type_out, w_out x h_out). */
int smol_hip_plan_job (uint32_t pixel_type_in, uint32_t width_in, uint32_t height_in,
                       uint32_t pixel_type_out, uint32_t width_out, uint32_t height_out,
                       uint32_t with_srgb, SmolHipJob *job);

/* Inclusive range of INPUT rows that output rows [first_row, first_row+n_rows)
 * depend on -- what a row-range shard has to hold (multi-GPU scatter). */
int smol_hip_input_row_range (const SmolHipJob *job, uint32_t first_row, uint32_t n_rows,
                              uint32_t *in_row_first, uint32_t *in_row_last);

/* Closed-form sampling parameters, for checking against the reference's tables:
 * bilinear sample j -> (source index, weight F in 0..256);
 * box k -> (first index, last index, F of the right edge, F of the left carry). */
void smol_hip_bilinear_sample (uint64_t frac0, uint64_t step, uint32_t dim_in, uint32_t j,
                               uint32_t *ofs, uint32_t *f);
void smol_hip_box_span (uint64_t step, uint32_t dim_in, uint32_t k,
                        uint32_t *ofs0, uint32_t *ofs1, uint32_t *f_right, uint32_t *f_prev);

const char *smol_hip_strerror (int err);

/* The kernel family the dispatcher gives these rows to (it depends on the job, on the alignment of the pointers
 * and strides and on the row range): "k_pow2_batch", "k_tile_batch", "k_lind_batch", ... For tools that count
 * which kernels a workload exercises; nothing is launched. */
const char *smol_hip_kernel_name (const SmolHipJob *job, const void *src, uint32_t rowstride_in,
                                  const void *dst, uint32_t rowstride_out, uint32_t first_row, uint32_t n_rows);

/* ---- device side --------------------------------------------------------- */

int smol_hip_device_count (void);

/* A plan binds a job to one device. It owns the intermediate image of the two-pass
 * box filter for the device-pointer entry (allocated once, by smol_hip_plan_prepare()
 * or on the first call that needs it); the host-pointer entries borrow one of the
 * device's staging lanes (streams + slabs) per call. */
int smol_hip_plan_create (const SmolHipJob *job, int device, SmolHipPlan **plan_out);
void smol_hip_plan_destroy (SmolHipPlan *plan);

/* Allocates whatever smol_hip_scale_rows_device() could need for this plan now, so that
 * later calls are purely asynchronous launches (legal under stream capture). Optional. */
int smol_hip_plan_prepare (SmolHipPlan *plan);
/* The same for callers that keep several source images in flight on one plan (frame N and N+1 on
 * alternating streams): allocates the intermediate images and ordering events for `images_in_flight'
 * (1..4) images now. Without it the second to fourth buffer are allocated on first use, and an
 * allocation synchronises the device and must not run while another thread captures a stream in
 * global capture mode. */
int smol_hip_plan_reserve (SmolHipPlan *plan, uint32_t images_in_flight);

/* Host-pointer entry (what smol_scale_batch() calls): copies the needed input
 * rows to the device, runs the kernels, copies rows [first_row, first_row+n_rows)
 * back to dest_host (row first_row at offset 0, rowstride_out apart; bytes
 * between rows are left untouched). Thread-safe on one plan. Synchronous.
 *
 * ANY host memory works, as with the reference (smolscale-generic.c:1613-1642 reads and writes
 * the caller's rows with plain loads and stores). Since round 5 the kernel works on page-locked
 * HOST rows across PCIe (the "mapped road"; no DMA engine): page-locked memory (smol_hip_host_alloc,
 * smol_hip_host_register, hipHostMalloc) is read and written IN PLACE by one launch over the whole
 * strip; PAGEABLE memory is never shown to the HIP runtime -- the library's copy threads move it
 * through a ring of page-locked buffers owned by the lane (in and out), in chunks of about 4 MB
 * whose kernels follow one another on one stream, so no page of the caller's heap is ever pinned or
 * touched by the GPU. Either side on its own. SMOL_HIP_MAPPED=0 brings back the road of rounds
 * 1-4: chunks of output rows through three side streams of a staging lane -- H2D of chunk k+1, the
 * kernels of chunk k and D2H of chunk k-1 concurrently, page-locked memory handed to the DMA engines
 * in place. A call that moves at most SMOL_HIP_ZEROCOPY_KB (default 256) KiB goes through ONE ring
 * slot whatever its memory is -- one launch and one wait per call, for callers that ask for their
 * rows a few at a time (the reference's own outer loop, smolscale.c:507-546). */
int smol_hip_scale_rows_host (SmolHipPlan *plan,
                              const void *pixels_in_host, uint32_t rowstride_in,
                              void *dest_host, uint32_t rowstride_out,
                              uint32_t first_row, uint32_t n_rows);

/* Callers that ask for their rows a few at a time (the reference's own outer loop is one row per step with a
 * two-row cache, smolscale.c:507-546, smolscale-generic.c:1648-1682): the second small call that continues where the
 * previous one ended computes a BAND of the rows that follow (about 2 MB of traffic) into a page-locked buffer the
 * plan keeps; later calls inside the band are a memcpy. Nothing in smolscale.h forbids rewriting pixels_in between
 * calls, so a call is served from the band only while the source rows its output rows read still hold the bytes
 * the band was computed from (a memcmp of two or three rows on a bilinear axis; calls that move more than 256 KiB
 * take the ordinary road). smol_hip_plan_resident_source (plan, 1) -- or SMOL_HIP_RESIDENT_SRC=1 in the
 * environment -- is the caller's promise that the source does not change while the plan lives: no comparison,
 * read-ahead for calls of any size, bands of 8 MB. SMOL_HIP_READAHEAD=0 turns read-ahead off.
 * The environment variable is a promise about EVERY plan of the process, including the contexts behind smolscale.h, whose
 * contract lets a caller rewrite pixels_in between calls: set it only for a process that never does; the per-plan call
 * is the safe form. Changing the setting of a plan drops everything computed ahead for it (bands in flight are waited
 * for), so a caller may take the promise back, rewrite its source and promise again. A call that finds the plan's
 * read-ahead state busy in another thread takes the ordinary road instead of waiting for it. */
int smol_hip_plan_resident_source (SmolHipPlan *plan, int resident);

/* The same over several devices: rows [first_row, first_row+n_rows) are cut into one
 * strip per plan (ceil (n_rows / n_plans) rows each -- the reference's worker-thread
 * rule, test.c:870-877), every strip goes through its own plan's device as above, all
 * strips in flight together. Rows are independent, so nothing is exchanged between
 * devices. plans [i] may name the same device more than once. With page-locked host
 * memory one host thread drives all devices asynchronously; with pageable memory the
 * strips are handed to the library's persistent worker threads (created once).
 * Every plan must have been created for the SAME job (one image, one geometry): a plan
 * whose job differs from plans [0]'s makes the call return SMOL_HIP_E_INVALID. */
int smol_hip_scale_rows_host_multi (SmolHipPlan *const *plans, uint32_t n_plans,
                                    const void *pixels_in_host, uint32_t rowstride_in,
                                    void *dest_host, uint32_t rowstride_out,
                                    uint32_t first_row, uint32_t n_rows);

/* As smol_hip_scale_rows_host(), but the source image already lives in the memory of GPU
 * `src_device' (device pointer): the slab of input rows travels GPU -> GPU over xGMI
 * (hipMemcpyPeerAsync) instead of host -> GPU. src_device may equal the plan's device. */
int smol_hip_scale_rows_peer (SmolHipPlan *plan, int src_device,
                              const void *pixels_in_dev, uint32_t rowstride_in,
                              void *dest_host, uint32_t rowstride_out,
                              uint32_t first_row, uint32_t n_rows);

/* Page-lock / unlock a caller-owned host range for the calls above (hipHostRegister).
 * Worth it when the same buffers serve many calls: locking costs about as much as one
 * transfer of the range. */
int smol_hip_host_register (void *host_ptr, uint64_t n_bytes);
int smol_hip_host_unregister (void *host_ptr);
/* Page-locked host memory from the runtime's own allocator (hipHostMalloc): the preferred way to
 * get buffers that take the zero-copy road. */
int smol_hip_host_alloc (uint64_t n_bytes, void **host_ptr_out);
int smol_hip_host_free (void *host_ptr);
/* 1 when EVERY byte of the range is page-locked (so that the host entries let the kernel work on it in place), 0
 * when any part of it is pageable or cannot be told (it then travels through the library's rings). The range is
 * walked allocation by allocation -- two registrations with a pageable gap between them are pageable. */
int smol_hip_host_range_is_pinned (const void *host_ptr, uint64_t n_bytes);

/* Device-pointer entry: input image and output strip already live in device
 * memory. Asynchronous on `hip_stream' (a hipStream_t, NULL = default stream): no
 * allocation and no synchronisation once the plan is prepared. Concurrent calls on one
 * plan -- from any threads, on any streams -- are fine: for disjoint row ranges of one source
 * image and for different source images alike. (The two-pass box filter keeps its intermediate
 * image in a ring of four buffers owned by the plan, one per image in flight; a fifth
 * concurrent image queues behind the oldest with hipStreamWaitEvent, and the second to
 * fourth buffers are allocated on first use. A captured call -- hipStreamBeginCapture --
 * always uses the first buffer and orders nothing: replays of such a graph must not overlap
 * other work on the same plan.) */
int smol_hip_scale_rows_device (SmolHipPlan *plan,
                                const void *pixels_in_dev, uint32_t rowstride_in,
                                void *dest_dev, uint32_t rowstride_out,
                                uint32_t first_row, uint32_t n_rows,
                                void *hip_stream);

/* Batched entry: n images (each with its own job, pointers and row range) in
 * as few launches as possible; one launch when all images use the same kernel
 * class. `images' is a HOST array. Asynchronous on hip_stream. */
typedef struct SmolHipBatch SmolHipBatch;
int smol_hip_batch_create (const SmolHipImage *images, uint32_t n_images, int device,
                           SmolHipBatch **batch_out);
int smol_hip_batch_launch (SmolHipBatch *batch, void *hip_stream);
/* enable != 0: submit the batch as a hipGraph holding one kernel node per image
 * (captured once, replayed by smol_hip_batch_launch) instead of one launch per
 * kernel class over a device-resident descriptor table. Same results; for A/B. */
int smol_hip_batch_use_graph (SmolHipBatch *batch, int enable);
/* Number of kernel launches one smol_hip_batch_launch() issues, and the name of
 * the dominant kernel (for matching against rocprofv3 --stats). */
int smol_hip_batch_info (const SmolHipBatch *batch, uint32_t *n_launches, const char **kernel_name);
void smol_hip_batch_destroy (SmolHipBatch *batch);

/* Timing helper for bench.py: `warmup' untimed then `iters' timed launches of
 * the batch on a private stream, bracketed by hipEvents recorded on that same
 * stream. Returns total milliseconds for the timed launches. */
int smol_hip_batch_time (SmolHipBatch *batch, uint32_t warmup, uint32_t iters, float *ms_total);
/* The same with one event between consecutive launches: ms_each [i] = duration of timed
 * launch i (for min / median). */
int smol_hip_batch_time_each (SmolHipBatch *batch, uint32_t warmup, uint32_t iters, float *ms_each);
/* Several batches of ONE device launched in rotation on one stream (launch i = batches [i % n_batches]): the same work
 * into several destination arenas, so that the bytes written between two writes of the same line exceed the device's
 * 256 MB Infinity Cache and every output byte has to leave it -- a batch launched again and again into one small arena
 * has its writes absorbed there (round 6: the headline's 128 MB of output per launch, 324 us; four arenas in rotation,
 * 334 us). `warmup' untimed then `iters' timed launches; *ms_total = the timed launches together; ms_each (may be NULL)
 * receives the `iters' single durations. */
int smol_hip_batch_time_rotating (SmolHipBatch *const *batches, uint32_t n_batches, uint32_t warmup, uint32_t iters,
                                  float *ms_total, float *ms_each);

/* Where to put a streaming job's source and destination arenas (round 6, profiles/r06_zone_map.txt). Device memory on
 * MI355X comes in three kinds -- 96 GB of each, by every sign the three ranks of its 12-high HBM3E stacks -- and a large
 * allocation is a patchwork of them in pieces of 4-64 GB. A kernel that READS one kind and WRITES another runs faster than
 * one that reads and writes the same kind -- the headline 325 against 336 us per launch into one destination arena that
 * the Infinity Cache absorbs, 326 against 350 us (7.5 %) once the output really goes to memory; a plain copy likewise --,
 * whichever kernel, stream or process; a kernel that reads two streams at once runs 7-9 % faster when both are of ONE kind.
 * Nothing in an address tells the kinds apart, so this call measures them: [base, base + bytes) is cut into windows of
 * `window' bytes (>= 512 MB, a multiple of 16; base 16-byte aligned; a trailing partial window is left out), pairs of
 * windows are read side by side (256 MB of each, a few launches), and zones [i] receives the kind of window i -- of its
 * first 256 MB: borders fall on multiples of 2-4 GB, not of a window -- as a small number: window 0 is kind 0, the first
 * window of another kind is kind 1, ... Two windows with DIFFERENT labels are a fast pair for source and destination;
 * windows with the SAME label are where to keep several sources that one launch reads. *n_kinds (may be NULL) = labels
 * handed out (3 on a region that spans them all; a window that straddles a border can come out as a kind of its own; at
 * most 8); *contrast (may be NULL) = the slowest pair against two halves of one window, 1.07-1.12 where more than one kind
 * is present (a disturbed measurement shows in it). The caller's memory is only read; no device memory is allocated.
 * n_zones >= bytes / window. About 1 s for 120 windows (0.15-0.5 s of it settling the clocks), 3 s for 480.
 * SMOL_HIP_ZONES_TRACE=1 in the environment prints every sweep's times on stderr.
 * Replaces nothing in the reference (its buffers are the caller's, smolscale.h:56-67): it is advice for callers who hand
 * this library device pointers, and what bench.py's placement step reports beside its own probes. */
int smol_hip_memory_zones (int device, const void *base, uint64_t bytes, uint64_t window,
                           uint8_t *zones, uint32_t n_zones, uint32_t *n_kinds, float *contrast);

/* A batch spread over several GPUs by image: image i belongs to devices [i % n_devices]
 * and its src/dst must be device pointers valid there (thumbnails: no image is split).
 * One descriptor table and one stream per listed device; launch fans out from the
 * calling thread (launches are asynchronous, no helper threads), sync waits for every
 * device. A device may be listed more than once. */
typedef struct SmolHipBatchMulti SmolHipBatchMulti;
int smol_hip_batch_create_multi (const SmolHipImage *images, uint32_t n_images,
                                 const int *devices, uint32_t n_devices,
                                 SmolHipBatchMulti **batch_out);
int smol_hip_batch_multi_launch (SmolHipBatchMulti *batch);
int smol_hip_batch_multi_sync (SmolHipBatchMulti *batch);
/* `warmup' + `iters' launches on every device concurrently; ms_per_device [d] = that
 * device's hipEvent time for its `iters' launches; returns after all devices finished. */
int smol_hip_batch_multi_time (SmolHipBatchMulti *batch, uint32_t warmup, uint32_t iters,
                               float *ms_per_device);
void smol_hip_batch_multi_destroy (SmolHipBatchMulti *batch);

/* Releases what the library holds per process: staging lanes (streams, slabs, events)
 * of every device and the worker threads. Slabs above 256 MB also go when the last plan
 * on a device is destroyed. Call with no smol_hip_* call in flight. */
void smol_hip_shutdown (void);

/* Device memory helpers so that hosts without a HIP binding (ctypes, cgo...)
 * can stage buffers. The copies are synchronous and ordered like hipMemcpy (behind what the
 * caller queued on blocking streams); pageable host memory travels through a bounce ring. */
int smol_hip_malloc (int device, uint64_t n_bytes, void **dev_ptr_out);
int smol_hip_free (int device, void *dev_ptr);
int smol_hip_memcpy_h2d (int device, void *dst_dev, const void *src_host, uint64_t n_bytes);
int smol_hip_memcpy_d2h (int device, void *dst_host, const void *src_dev, uint64_t n_bytes);
int smol_hip_memset (int device, void *dst_dev, int value, uint64_t n_bytes);
int smol_hip_device_synchronize (int device);

#ifdef __cplusplus
}
#endif

#endif /* SMOLSCALE_AMD_SMOL_HIP_H */
