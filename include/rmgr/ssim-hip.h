/*
 * rmgr/ssim-hip.h -- the C ABI of the MI355X (gfx950) SSIM engine.
 *
 * This is the drop-in boundary: plain C, pointers and sizes only, no HIP / torch / C++ types.
 * <rmgr/ssim.h>'s entry points are thin C++98 wrappers over it (ssim_amd/csrc/ssim_dropin.cpp),
 * and any other host (ctypes, cgo, JNI ...) binds these symbols directly -- see INTEGRATION.md.
 *
 * Each function names the reference interface it stands in for (file:line in romigrou/ssim):
 *
 *   rmgr_ssim_hip_compute_ssim_host    rmgr::ssim::compute_ssim           src/ssim.cpp:933-1106
 *   rmgr_ssim_hip_compute_ssim_device  same, images/map already in HBM    src/ssim.cpp:933-1106
 *   rmgr_ssim_hip_compute_ssim_batch_host  a caller's loop over host pairs  sample/rmgr-ssim-sample.cpp:84-95
 *   rmgr_ssim_hip_compute_ssim_batch_host_devices  the same over several GPUs   src/ssim.cpp:1048-1088 (thread-pool dispatch)
 *   rmgr_ssim_hip_enqueue_batch        the caller-side loop over pairs    src/ssim-cli.cpp:197-210
 *                                      + per-thread fp64 partials         src/ssim.cpp:902-926
 *   rmgr_ssim_hip_enqueue_rows / _reduce_cells   one image's tile jobs split over workers   src/ssim.cpp:1048-1100
 *   rmgr_ssim_hip_finalize             the final mean                     src/ssim.cpp:1090-1103
 *   rmgr_ssim_hip_compute_ssim_channels_host   the per-channel caller loop       src/ssim-cli.cpp:197-210, sample/rmgr-ssim-sample.cpp:82-101
 *   rmgr_ssim_hip_compute_ssim_luminance_host  RGB -> BT.601 Y, then SSIM        src/ssim-cli.cpp:145-195
 *   rmgr_ssim_hip_luminance_device             the conversion loop alone          src/ssim-cli.cpp:158-186
 *   rmgr_ssim_hip_set_mode             select_impl() / RMGR_SSIM_USE_DOUBLE   src/ssim.cpp:808-896, src/ssim_internal.h:26-37
 *
 * All functions return 0 or an errno value (EINVAL, ENOMEM, ECHILD = a HIP call failed,
 * ENODEV = no gfx950 device / extension not usable), exactly like the reference's API; the multi-GPU
 * exchange (rmgr_ssim_hip_comm_*) adds ENOSYS = no librccl and ETIMEDOUT = a peer rank did not arrive in time.
 */
#ifndef RMGR_SSIM_HIP_H
#define RMGR_SSIM_HIP_H

#include <rmgr/ssim.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Arithmetic of the blur + SSIM stage. */
#define RMGR_SSIM_HIP_MODE_EXACT   0  /* operation order of the reference's FMA path (src/ssim_fma.cpp:196-257,
                                         src/ssim_avx.cpp:342-352): bit-identical per-pixel results. Default. */
#define RMGR_SSIM_HIP_MODE_FAST    1  /* the three E[.] planes in the reference's exact operation order (bit-identical planes), the two
                                         mu planes by a separable 11+11 fp32 blur: not bit-identical, but inside the reference's
                                         documented single-precision tolerance RELATIVE TO ITS FMA PATH (global 1.5e-6, per pixel
                                         6.3e-4) on all five of the reference's test image sets, with >= 30 % / >= 60 % margin.  (On large flat
                                         areas every pixel's rounding error has the same sign and the GLOBAL value can differ more --
                                         DESIGN.md section 2; only modes 0 and 3 are guarantees) */
#define RMGR_SSIM_HIP_MODE_DOUBLE  2  /* RMGR_SSIM_USE_DOUBLE semantics: fp64 internals, true double kernel (tests/ssim_naive.h) */
#define RMGR_SSIM_HIP_MODE_UNFUSED 3  /* operation order of the reference's AVX/SSE/generic paths (mul and add rounded separately) */
#define RMGR_SSIM_HIP_MODE_SEPARABLE 4 /* every plane by the separable 11+11 fp32 blur (four planes, centred pixels): the fastest mode and
                                         closer to the exact value than the reference's own fp32 paths (<= 2e-4 per pixel), inside the
                                         reference's TEST tolerances against its double oracle (2e-6 / 1e-3) -- but not correlated with
                                         the reference's rounding, hence NOT guaranteed within 6.3e-4 of its FMA path per pixel */

/* Version of THIS header's interface (the rmgr_ssim_hip_* functions and structs), independent of the reference API's
 * 2.1.0 that rmgr_ssim_get_version() reports.  Bumped whenever a struct layout, a function signature or the meaning of a
 * constant changes; rmgr_ssim_hip_get_abi_version() returns the value the LIBRARY was built with, so a client can
 * refuse a mismatch at start-up.  History (INTEGRATION.md has the migration notes):
 *   3  round 3: MODE_FAST (1) became the hybrid, the all-separable arithmetic moved to MODE_SEPARABLE (4); Plan grew by two fields
 *   4  round 4: Plan carries structSize (first field) and the cell grid; rmgr_ssim_hip_comm_* calls are bounded by a deadline
 *      (ETIMEDOUT), comm_rank_count / comm_describe added; row-band entry points (enqueue_rows, reduce_cells) added
 *   5  round 5: MODE_FAST (1) and MODE_SEPARABLE (4) form their quotient as n * rcp(d): their values moved by <= 3 ulp (contracts unchanged); ctx == NULL calls
 *      run on a pool of default contexts and no longer serialise; Plan grew by balancedChunks / balancedChunkRows (harmless: structSize);
 *      the deadline of synchronize / destroy applies per queued all-reduce; get_default_pool, get_kernel_source_id added
 *   6  round 6: additions only -- probe_valu, get_profile_clock, tune / get_tuned / set_tuned / clear_tuned, trim / trim_default_pool / get_default_pool_memory / get_memory_info; the default contexts
 *      release staging above $RMGR_SSIM_HIP_POOL_RETAIN_MB when a call ends; a threadPool with a dispatch function IS called (one job per row band, ECHILD
 *      when it fails); Plan: balancedInterleave appended (harmless: structSize), tuning variants 7 and 100 + T; set_mode / get_mode(NULL) no longer wait for a lease */
#define RMGR_SSIM_HIP_ABI_VERSION 6
rmgr_int32_t rmgr_ssim_hip_get_abi_version(void) RMGR_NOEXCEPT;

/* sha256 (hex) of the kernel source this library's device code was compiled from ("unknown" when it was not built by the Makefile).
 * Measurements tied to one version of the kernels -- profiles/traffic.json -- record it, and bench.py quotes them only for that version. */
const char* rmgr_ssim_hip_get_kernel_source_id(void) RMGR_NOEXCEPT;

/* The default contexts of the ctx == NULL entry points (see rmgr_ssim_hip_compute_ssim_host): how many exist right now and how many
 * calls may be in flight at a time.  Either pointer may be NULL.  Creates nothing. */
rmgr_int32_t rmgr_ssim_hip_get_default_pool(rmgr_int32_t* contexts, rmgr_int32_t* limit) RMGR_NOEXCEPT;

/* Memory policy of the default contexts.  A context's staging -- device copies of the images and the map, cell partials, descriptor tables,
 * pinned host mirrors and bounce buffers -- is grow-only WHILE a call runs, so that a caller looping over frames of one size allocates once.  The
 * reference keeps nothing past the call (src/ssim.cpp:1048-1088: one alloc / dealloc pair inside compute_ssim, include/rmgr/ssim.h:505-525), so:
 *   - when a ctx == NULL call ends and its context holds more than $RMGR_SSIM_HIP_POOL_RETAIN_MB (per context, device + pinned; default 256;
 *     0: keep nothing; negative: no cap) everything is released before the context is leased again: an 8192^2 + map call (134 MB + 268 MB of
 *     staging) leaves nothing behind, a 4096^2 + map loop (100 MB) keeps its buffers;
 *   - rmgr_ssim_hip_trim_default_pool() releases the staging of every default context that is not inside a call right now (the contexts themselves --
 *     stream, events -- stay: the next call re-grows what it needs; contexts in use are skipped);
 *   - rmgr_ssim_hip_get_default_pool_memory() reports what the default contexts held when their last call ended (device bytes, pinned host bytes)
 *     and the cap in force (bytes; UINT64_MAX: none).  Any pointer may be NULL.  Creates nothing.
 * rmgr_ssim_hip_trim(ctx) does the same for a caller-owned context (which has no cap: its owner decides; it must not be in use by another
 * thread; queued work is waited for first); ctx == NULL: the default pool.  Results never depend on any of this. */
rmgr_int32_t rmgr_ssim_hip_trim_default_pool(void) RMGR_NOEXCEPT;
rmgr_int32_t rmgr_ssim_hip_get_default_pool_memory(rmgr_uint64_t* deviceBytes, rmgr_uint64_t* pinnedBytes, rmgr_uint64_t* retainCapBytes) RMGR_NOEXCEPT;

/* An engine instance: one device, one stream, its own grow-only scratch.  A context may be used by one
 * host thread at a time (create one per thread, or serialise); ctx == NULL entry points lease one of the
 * process-wide default contexts per call (rmgr_ssim_hip_compute_ssim_host) and may be called from any number of threads. */
typedef struct rmgr_ssim_hip_Context_ rmgr_ssim_hip_Context;

/* Number of usable HIP devices (0 when none; never fails). */
rmgr_int32_t rmgr_ssim_hip_get_device_count(rmgr_int32_t* count) RMGR_NOEXCEPT;

/* Creates an engine bound to `device`.  `stream` is a hipStream_t passed as void* (NULL: the
 * context creates and owns a non-blocking stream).  All work of the context is ordered on it. */
rmgr_int32_t rmgr_ssim_hip_create(rmgr_ssim_hip_Context** ctx, rmgr_int32_t device, void* stream) RMGR_NOEXCEPT;
rmgr_int32_t rmgr_ssim_hip_destroy(rmgr_ssim_hip_Context* ctx) RMGR_NOEXCEPT;

rmgr_int32_t rmgr_ssim_hip_trim(rmgr_ssim_hip_Context* ctx) RMGR_NOEXCEPT;

/* Free and total memory of the context's device in bytes (hipMemGetInfo), for hosts without a HIP runtime of their own; ctx == NULL: the device
 * of the default contexts ($RMGR_SSIM_HIP_DEVICE).  Either pointer may be NULL.  ENODEV without a device. */
rmgr_int32_t rmgr_ssim_hip_get_memory_info(const rmgr_ssim_hip_Context* ctx, rmgr_uint64_t* freeBytes, rmgr_uint64_t* totalBytes) RMGR_NOEXCEPT;

/* ctx == NULL (set and get): the arithmetic mode of the process-wide default contexts the unchanged rmgr_ssim_compute_ssim() runs on
 * (a property of their pool, applied when a call leases one: calls in flight keep theirs; neither call waits for a lease or creates a context when
 * one exists already -- only on an empty pool is the first context created, so that a machine without a device answers ENODEV).  This is what
 * rmgr::ssim::select_impl() calls (src/ssim.cpp:808-896). */
rmgr_int32_t rmgr_ssim_hip_set_mode(rmgr_ssim_hip_Context* ctx, rmgr_int32_t mode) RMGR_NOEXCEPT;
rmgr_int32_t rmgr_ssim_hip_get_mode(const rmgr_ssim_hip_Context* ctx, rmgr_int32_t* mode) RMGR_NOEXCEPT;

/* Tuning knobs (0 = library default): rows of the image each wavefront strip covers, and the kernel variant --
 *   1  one column per lane (64-column strips);            2  two columns per lane, row sums in the blur phase;
 *   3  two columns per lane with the bit-exact modes' (a,b) row sums formed a phase early (EARLY);
 *   6  the balanced schedule of the two-column kernel (launches without a map; modes 0, 3, 1) with the library's interleave of the images in
 *      the chunk list; 7: without any interleave (round 5's list); 100 + T: T images interleaved (measurement aids, tools/phase_ab.sh);
 * the default picks by launch size (rmgr_ssim_hip_get_plan reports what it picked; rmgr_ssim_hip_tune measures the candidates on the device).
 * Results do not depend on either (tests/test_gpu_parity.py, tests/test_gpu_pipeline.py check). */
rmgr_int32_t rmgr_ssim_hip_set_tuning(rmgr_ssim_hip_Context* ctx, rmgr_int32_t stripRows, rmgr_int32_t variant) RMGR_NOEXCEPT;

/* How a launch of `count` width x height pairs is cut into wavefront strips under the context's mode and
 * tuning (ctx may be NULL: default mode, default tuning, a 256-CU device).  Pure host arithmetic: no device
 * is touched.  The reference's counterpart is its 256x64 tile grid (src/ssim.cpp:1026-1028).
 * The caller sets plan->structSize = sizeof(rmgr_ssim_hip_Plan) BEFORE the call; the library fills the fields that fit
 * in that many bytes and nothing beyond them (EINVAL below RMGR_SSIM_HIP_PLAN_MIN_SIZE), so the struct can grow
 * without overrunning the storage of a client compiled against an earlier header. */
typedef struct rmgr_ssim_hip_Plan
{
    rmgr_uint32_t structSize;        /* in: sizeof(rmgr_ssim_hip_Plan) as the CALLER was compiled */
    rmgr_uint32_t stripWidth;        /* output columns per wavefront: 128 (two per lane) or 64 (one per lane: fp64 mode, tiny launches, tuning variant 1) */
    rmgr_uint32_t stripRows;         /* output rows per wavefront STRIP.  When balancedChunks > 0 this and the next four fields describe the strips a launch WITH a map */
    rmgr_uint32_t stripsX, stripsY;  /* strips per image                  of these pairs runs; a launch without one runs balancedChunks wavefronts of balancedChunkRows rows, */
    rmgr_uint32_t wavefronts;        /* stripsX * stripsY * count         in the EARLY form for modes 0 and 3 (128 x 1080p: 5760 strips with a map, 2041 chunks without) */
    /* -- RMGR_SSIM_HIP_PLAN_MIN_SIZE ends here -- */
    rmgr_uint32_t waveSlots;         /* wavefronts the device holds at a time with this kernel (SIMDs x waves per SIMD) */
    rmgr_uint32_t earlyRowSums;      /* 1: the bit-exact two-column kernel's STRIPS run in the EARLY form (launches of <= 3 x waveSlots wavefronts) */
    rmgr_uint32_t cellRows;          /* rows of a reduction cell (64 columns x cellRows rows): 8, or 32 for images of >= 2048 rows */
    rmgr_uint32_t cellsX, cellsY;    /* the image's grid of reduction cells: cellsX * cellsY fp64 partials per image (rmgr_ssim_hip_enqueue_rows) */
    rmgr_uint32_t balancedChunks;    /* > 0: a launch of these pairs WITHOUT a map runs the balanced schedule of the two-column kernel (modes 0, 3, 1) -- this many */
    rmgr_uint32_t balancedChunkRows; /*      wavefronts, each walking this many rows of the launch's flattened [images][strip column][row] list -- instead of */
                                     /*      the strips above (same results bit for bit; scheduling only); 0: the strips */
    rmgr_uint32_t balancedInterleave;/* round 6: images interleaved column by column in that list (1: none): neighbouring strip columns of an image stay in step */
} rmgr_ssim_hip_Plan;
#define RMGR_SSIM_HIP_PLAN_MIN_SIZE 24u
rmgr_int32_t rmgr_ssim_hip_get_plan(const rmgr_ssim_hip_Context* ctx, rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint32_t count, rmgr_ssim_hip_Plan* plan) RMGR_NOEXCEPT;

/*
 * The plan of one launch shape, MEASURED.  rmgr_ssim_hip_get_plan reports the library's untuned default -- a model of how strips and chunks pack
 * onto the device, fitted on 256-CU MI355X boxes; rmgr_ssim_hip_tune times the candidates that model chooses between (the default; the strips at
 * the default height with the row sums in the blur phase / a phase early; half and twice the strip height; the balanced schedule where it
 * exists; the one-column kernel for small launches) on the context's own device, under the context's arithmetic mode, on synthetic pairs of
 * the shape it allocates and frees itself (distinct images up to ~1.5 GB; withMap: dense float maps; where the balanced schedule exists also the strips at its chunk
 * height), candidates interleaved over three rounds,
 * and keeps the winner for this context's later launches of exactly that shape (width, height, count, map or not, mode) while the context is
 * on its default tuning (set_tuning(ctx, 0, 0)); a winner has to beat the default by more than 0.5 %.  Results never depend on the choice.
 * Blocking (a few dozen launches of the shape).  The reference's counterpart is a caller choosing its thread count (include/rmgr/ssim.h:528-533).
 * result (may be NULL): the caller sets structSize; candidateXxx[0] is the default plan.  rmgr_ssim_hip_clear_tuned forgets every choice.
 */
#define RMGR_SSIM_HIP_TUNE_MAX_CANDIDATES 8
typedef struct rmgr_ssim_hip_TuneResult
{
    rmgr_uint32_t structSize;        /* in: sizeof(rmgr_ssim_hip_TuneResult) as the CALLER was compiled */
    rmgr_uint32_t candidates;        /* plans timed (<= RMGR_SSIM_HIP_TUNE_MAX_CANDIDATES) */
    rmgr_int32_t  bestVariant;       /* the winner as rmgr_ssim_hip_set_tuning arguments (0 / 0: the default stays) */
    rmgr_uint32_t bestStripRows;
    double        defaultMs, bestMs; /* kernel time of the default plan and of the winner: median over rounds of the mean of three launches */
    /* -- RMGR_SSIM_HIP_TUNE_RESULT_MIN_SIZE ends here -- */
    rmgr_int32_t  candidateVariant[RMGR_SSIM_HIP_TUNE_MAX_CANDIDATES];
    rmgr_uint32_t candidateStripRows[RMGR_SSIM_HIP_TUNE_MAX_CANDIDATES];
    double        candidateMs[RMGR_SSIM_HIP_TUNE_MAX_CANDIDATES];
} rmgr_ssim_hip_TuneResult;
#define RMGR_SSIM_HIP_TUNE_RESULT_MIN_SIZE 32u
rmgr_int32_t rmgr_ssim_hip_tune(rmgr_ssim_hip_Context* ctx, rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint32_t count, rmgr_int32_t withMap,
                                rmgr_ssim_hip_TuneResult* result) RMGR_NOEXCEPT;
rmgr_int32_t rmgr_ssim_hip_clear_tuned(rmgr_ssim_hip_Context* ctx) RMGR_NOEXCEPT;
/* The measured choices a context holds, for hosts that tune once per machine and keep the result (a serving process that restores them at start-up pays no tuning):
 * rmgr_ssim_hip_get_tuned reads entry `index` (0, 1, ... until ENOENT) -- the launch shape (width, height, count, withMap), the arithmetic mode it was measured
 * under and the winner as rmgr_ssim_hip_set_tuning arguments; rmgr_ssim_hip_set_tuned installs (or replaces) an entry without measuring anything, for the context's CURRENT
 * mode (EINVAL for a shape with a zero dimension or count, a negative variant, or strip rows and variant both 0: that is the default, use clear_tuned).  Like every tuning it
 * changes scheduling only. */
typedef struct rmgr_ssim_hip_TunedEntry
{
    rmgr_uint32_t width, height, count;
    rmgr_int32_t  withMap, mode, variant;
    rmgr_uint32_t stripRows;
} rmgr_ssim_hip_TunedEntry;
rmgr_int32_t rmgr_ssim_hip_get_tuned(const rmgr_ssim_hip_Context* ctx, rmgr_uint32_t index, rmgr_ssim_hip_TunedEntry* entry) RMGR_NOEXCEPT;
rmgr_int32_t rmgr_ssim_hip_set_tuned(rmgr_ssim_hip_Context* ctx, rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint32_t count, rmgr_int32_t withMap,
                                     rmgr_int32_t variant, rmgr_uint32_t stripRows) RMGR_NOEXCEPT;

/*
 * compute_ssim() on HOST pointers: stages both images to HBM, runs the kernels, copies the map
 * back (any ssimStep/ssimStride), returns the global SSIM.  Validation and return codes are the
 * reference's (src/ssim.cpp:962-978).  ctx may be NULL: the call then runs on one of the process-wide DEFAULT contexts
 * on device 0 (or $RMGR_SSIM_HIP_DEVICE), leased for the duration of the call.  Like the reference's function
 * (re-entrant, no global state: src/ssim.cpp:933-1106) concurrent callers run side by side -- each on a context of its
 * own (stream, staging buffers, pinned memory), one caller's copy-in under another's kernel and a third's map on its way
 * back -- up to $RMGR_SSIM_HIP_POOL calls at a time (default 4; 1 serialises them as rounds 1-4 did); further callers
 * wait for a lease.  Contexts are created on demand: a single-threaded process has one.
 * A large pair with a map is processed in row bands -- copy-in of band k+1, kernel on band k and the copy-back of
 * band k-1's map rows (by a short-lived helper thread, straight into ssimMap) overlap; the results are bit-identical
 * to the one-launch computation.  $RMGR_SSIM_HIP_BANDS overrides the band count (1: no overlap) and sends a large pair through
 * the banded path whatever its size and whether or not a map is wanted (measurement aid).
 * The calling thread's current HIP device is left as it was (true of every function in this header).
 */
rmgr_int32_t rmgr_ssim_hip_compute_ssim_host(rmgr_ssim_hip_Context* ctx, float* ssim, const rmgr_ssim_Params* params,
                                             const rmgr_ssim_ThreadPool* threadPool) RMGR_NOEXCEPT;

/*
 * Same computation with imgA.topLeft, imgB.topLeft and ssimMap being DEVICE pointers (step/stride
 * semantics unchanged).  `ssim` is a host pointer; the call returns after the result is there.
 */
rmgr_int32_t rmgr_ssim_hip_compute_ssim_device(rmgr_ssim_hip_Context* ctx, float* ssim, const rmgr_ssim_Params* params) RMGR_NOEXCEPT;

/*
 * Asynchronous batch: `count` pairs of identical width/height, all pointers device-resident.
 * One launch covers the whole batch; image i's fp64 sum of per-pixel SSIM values is written to
 * sumsDevice[i] (device memory, count doubles).  The reduction is organised in cells at fixed image positions and
 * summed in a fixed order, so the value is bit-identical however a batch is split across calls, contexts or GPUs
 * and whatever the tuning.  Different batches may be enqueued back to back: the descriptor tables are kept in a
 * small ring, nothing waits for the stream.  Returns once enqueued.
 */
rmgr_int32_t rmgr_ssim_hip_enqueue_batch(rmgr_ssim_hip_Context* ctx, rmgr_uint32_t count, const rmgr_ssim_Params* params,
                                         double* sumsDevice) RMGR_NOEXCEPT;

/*
 * ONE image pair cut into ROW BANDS -- for an image too large, or too urgent, for one GPU ("single huge
 * image across GPUs"; the reference's counterpart is its tile grid walked by several threads, src/ssim.cpp:1048-1100).
 * Everything device-resident and asynchronous on the context's stream.
 *   rmgr_ssim_hip_enqueue_rows   computes output rows [yBegin, yBegin + yRows) of the pair (clipped to the image): the map
 *       rows of the band, if params->ssimMap is set, and the band's reduction-cell partials -- 64 columns x cellRows rows
 *       each, rmgr_ssim_hip_get_plan() reports cellRows / cellsX / cellsY -- into cellsDevice[cellY * cellsX + cellX], an
 *       array of cellsX * cellsY doubles the caller ZEROED; cells outside the band are not touched.  yBegin must be a
 *       multiple of cellRows and the band must end on one or at the last row (EINVAL otherwise).  The image pointers
 *       describe the WHOLE image (topLeft = row 0), but only source rows yBegin - 5 ... yEnd + 4 (clamped to the image)
 *       are read: a GPU that owns a band needs just those rows resident.
 *   rmgr_ssim_hip_reduce_cells   sums `count` images' complete cell arrays ([image][cellY][cellX]) into sumsDevice[image]
 *       in the fixed order every launch of this library uses.
 * Bands may be computed by different contexts or GPUs into separate zeroed arrays; adding the arrays element-wise
 * (rmgr_ssim_hip_comm_allreduce_sums on the cell array: every cell is non-zero on exactly one rank, and adding zeros is
 * exact) and reducing the result gives the SAME BITS as one launch over the whole image on one GPU.
 */
rmgr_int32_t rmgr_ssim_hip_enqueue_rows(rmgr_ssim_hip_Context* ctx, const rmgr_ssim_Params* params, rmgr_uint32_t yBegin, rmgr_uint32_t yRows,
                                        double* cellsDevice) RMGR_NOEXCEPT;
rmgr_int32_t rmgr_ssim_hip_reduce_cells(rmgr_ssim_hip_Context* ctx, rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint32_t count,
                                        const double* cellsDevice, double* sumsDevice) RMGR_NOEXCEPT;

/*
 * A batch of HOST image pairs (identical width/height, global SSIM only: every ssimMap must be NULL): what a
 * caller looping rmgr_ssim_compute_ssim() over frames does (sample/rmgr-ssim-sample.cpp:84-95, src/ssim-cli.cpp:
 * 197-210), with the staging pipelined -- pairs are copied to the GPU in chunks on a second stream while the kernels
 * of the previous chunk run, small pairs are gathered in pinned memory and sent with one DMA per chunk.  ssim[i]
 * is bit-identical to the single-pair call on pair i.  ctx may be NULL (process-wide default context).  Blocking.
 */
rmgr_int32_t rmgr_ssim_hip_compute_ssim_batch_host(rmgr_ssim_hip_Context* ctx, rmgr_uint32_t count, const rmgr_ssim_Params* params,
                                                   float* ssim) RMGR_NOEXCEPT;

/*
 * The same batch of HOST pairs sharded BY IMAGE over several devices from ONE process ("one process
 * x 8 devices"): contiguous blocks of the batch -- the first count % n one pair longer -- one per entry of `devices`
 * (NULL: every visible device, deviceCount ignored), each on its own worker thread and engine context (created on first
 * use, cached), each through the pipelined staging of rmgr_ssim_hip_compute_ssim_batch_host.  No image data crosses
 * devices and, in one address space, there is no exchange step either: every ssim[i] is written by the worker that owns
 * pair i and is bit-identical to the single-device result, whatever the device list.  A device may be listed more than
 * once (several contexts on one GPU).  `mode` is the arithmetic of every worker.  Stands in for the reference's
 * thread-pool dispatch (src/ssim.cpp:1048-1088, src/ssim-openmp.c:26-47) at batch granularity.  Blocking; one call at a
 * time per process.  EINVAL for a device index that is not visible, ENODEV without devices.
 */
rmgr_int32_t rmgr_ssim_hip_compute_ssim_batch_host_devices(const rmgr_int32_t* devices, rmgr_uint32_t deviceCount, rmgr_int32_t mode,
                                                           rmgr_uint32_t count, const rmgr_ssim_Params* params, float* ssim) RMGR_NOEXCEPT;

/* ssim[i] = float(sums[i] / double(width*height)) with the reference's 32-bit product (src/ssim.cpp:1102).  Host arrays. */
rmgr_int32_t rmgr_ssim_hip_finalize(rmgr_uint32_t count, const double* sums, rmgr_uint32_t width, rmgr_uint32_t height, float* ssim) RMGR_NOEXCEPT;

/*
 * All channels of one interleaved pair (host pointers, `channelCount` bytes per pixel, rows
 * `strideA` / `strideB` bytes apart) in ONE staging copy and ONE launch: ssim[c] receives channel c's
 * global SSIM.  ssimMap (or NULL) is an interleaved float map, channelCount floats per pixel, rows
 * width*channelCount floats apart -- the layout rmgr-ssim writes (src/ssim-cli.cpp:108-127).
 */
rmgr_int32_t rmgr_ssim_hip_compute_ssim_channels_host(rmgr_ssim_hip_Context* ctx, float* ssim,
                                                      const rmgr_uint8_t* imgA, ptrdiff_t strideA, const rmgr_uint8_t* imgB, ptrdiff_t strideB,
                                                      rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint32_t channelCount, float* ssimMap) RMGR_NOEXCEPT;

/*
 * SSIM of the BT.601 luminance of two interleaved images with >= 3 channels:
 * Y = (19595 R + 38470 G + 7471 B + 32768) / 65536 in integers (src/ssim-cli.cpp:158-186), computed on
 * the GPU from one staging copy.  ssimMap (or NULL): dense width x height floats.
 */
rmgr_int32_t rmgr_ssim_hip_compute_ssim_luminance_host(rmgr_ssim_hip_Context* ctx, float* ssim,
                                                       const rmgr_uint8_t* imgA, ptrdiff_t strideA, const rmgr_uint8_t* imgB, ptrdiff_t strideB,
                                                       rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint32_t channelCount, float* ssimMap) RMGR_NOEXCEPT;

/* The conversion alone, device to device (asynchronous on the context's stream): dstY[x + y*dstStride]
 * from src[x*srcStep + y*srcStride + {0,1,2}].  srcStep >= 3. */
rmgr_int32_t rmgr_ssim_hip_luminance_device(rmgr_ssim_hip_Context* ctx, rmgr_uint8_t* dstY, ptrdiff_t dstStride,
                                            const rmgr_uint8_t* src, ptrdiff_t srcStep, ptrdiff_t srcStride,
                                            rmgr_uint32_t width, rmgr_uint32_t height) RMGR_NOEXCEPT;

/*
 * Multi-GPU exchange without any other runtime: one process per GPU, images sharded by rank (no image
 * data crosses GPUs), every rank enqueues its shard into ITS slice of a zero-initialised device vector
 * of per-image fp64 sums, then all ranks call comm_allreduce_sums on the whole vector: one RCCL
 * all-reduce(sum, fp64) over xGMI.  Adding zeros is exact, so the result does not depend on the GPU
 * count.  This is the GPU-era form of the reference's per-thread partials + final loop
 * (src/ssim.cpp:902-926, :1094-1100).  librccl is loaded on first use ($RMGR_SSIM_HIP_RCCL_LIB, else the
 * copy already in the process or the system's); ENOSYS if it is absent.
 *   rank 0: comm_get_unique_id(id); ship the 128 bytes to the other ranks by any means (file, socket, MPI)
 *   all:    comm_init(ctx, id, rankCount, rank);  ...  comm_allreduce_sums(ctx, sumsDevice, count);
 *
 * Bounded failure.  Where the reference reports a failed worker as ECHILD (src/ssim.cpp:1094-1097), a rank that never
 * arrives is reported here as ETIMEDOUT after $RMGR_SSIM_HIP_COMM_TIMEOUT_S seconds (default 30): get_unique_id and
 * comm_init (library load, bootstrap, rendezvous), the enqueue inside comm_allreduce_sums, and every wait for a QUEUED
 * all-reduce -- rmgr_ssim_hip_synchronize(), comm_destroy and rmgr_ssim_hip_destroy on a context with all-reduces outstanding --
 * are bounded by it.  The deadline of a queued all-reduce runs from the moment its turn on the stream has come, not from the call:
 * kernels queued before or after it may take as long as they take (each all-reduce sits between two events; the wait polls them,
 * sleeping 50 us ... 1 ms between polls).  Past the deadline the communicator is aborted (ncclCommAbort), which releases the
 * kernel that waits for the missing peers; the wait for that kernel to leave is bounded by one more such interval.  A library
 * without ncclCommAbort cannot release it: ETIMEDOUT is returned and the stream is left as it is.  ncclCommDestroy of a BLOCKING
 * communicator ($RMGR_SSIM_HIP_COMM_BLOCKING=1) waits for the peers inside RCCL and is not bounded; the default (non-blocking)
 * communicator's teardown is.  After ETIMEDOUT the context is back in its single-GPU state -- it still computes, and
 * comm_init may be called again (tests/test_gpu_zz_rccl.py does both).  How: library load, ncclGetUniqueId and the
 * communicator's creation run on a helper thread the caller waits for with a timeout; the communicator is requested
 * non-blocking (ncclCommInitRankConfig, blocking = 0) and, if the deadline passes while its rendezvous is in progress,
 * aborted (ncclCommAbort) by the helper.  On the RCCL builds of this image (2.26.6 in the PyTorch wheel, 2.27.7 in
 * /opt/rocm) that call carries the rendezvous out before it returns, blocking = 0 notwithstanding (measured:
 * profiles/r04_final_rccl_selftest.txt), so what bounds comm_init there is the helper's timeout: the helper stays parked
 * inside RCCL and is abandoned (one idle thread; nothing it touches lives on the caller's stack).  Caveats of an abandoned helper: it is
 * still inside librccl when the process exits or this library is unloaded (exit from main is safe -- the thread is never joined and owns
 * its data --, dlclose() of this library while it exists is not); and while a FIRST use is still loading librccl on another thread,
 * comm_describe / comm_allreduce_sums / comm_destroy report "not loadable" / ENOSYS instead of waiting for that load.  Helpers never read
 * the environment: $RMGR_SSIM_HIP_RCCL_LIB, _COMM_DEBUG and _COMM_TIMEOUT_S are read on the calling thread of each entry point.
 * $RMGR_SSIM_HIP_COMM_BLOCKING=1 asks for a plain blocking communicator; $RMGR_SSIM_HIP_COMM_DEBUG=1 logs the helper's
 * steps on stderr.
 */
#define RMGR_SSIM_HIP_COMM_ID_BYTES 128
rmgr_int32_t rmgr_ssim_hip_comm_get_unique_id(unsigned char id[RMGR_SSIM_HIP_COMM_ID_BYTES]) RMGR_NOEXCEPT;
rmgr_int32_t rmgr_ssim_hip_comm_init(rmgr_ssim_hip_Context* ctx, const unsigned char id[RMGR_SSIM_HIP_COMM_ID_BYTES],
                                     rmgr_int32_t rankCount, rmgr_int32_t rank) RMGR_NOEXCEPT;
rmgr_int32_t rmgr_ssim_hip_comm_allreduce_sums(rmgr_ssim_hip_Context* ctx, double* sumsDevice, rmgr_uint32_t count) RMGR_NOEXCEPT;
rmgr_int32_t rmgr_ssim_hip_comm_destroy(rmgr_ssim_hip_Context* ctx) RMGR_NOEXCEPT;
/* Ranks RCCL itself counts in the context's communicator (ncclCommCount); 0 when it has none. */
rmgr_int32_t rmgr_ssim_hip_comm_rank_count(const rmgr_ssim_hip_Context* ctx, rmgr_int32_t* rankCount) RMGR_NOEXCEPT;
/* Which RCCL was loaded (version, path), whether non-blocking init is available, the deadline in force; static storage. */
const char* rmgr_ssim_hip_comm_describe(void) RMGR_NOEXCEPT;

/* Blocks until everything enqueued on the context's stream has finished. */
rmgr_int32_t rmgr_ssim_hip_synchronize(rmgr_ssim_hip_Context* ctx) RMGR_NOEXCEPT;

/* Device memory for hosts that have no HIP runtime of their own. */
rmgr_int32_t rmgr_ssim_hip_malloc(rmgr_ssim_hip_Context* ctx, void** devicePtr, size_t size) RMGR_NOEXCEPT;
rmgr_int32_t rmgr_ssim_hip_free(rmgr_ssim_hip_Context* ctx, void* devicePtr) RMGR_NOEXCEPT;
rmgr_int32_t rmgr_ssim_hip_memcpy_h2d(rmgr_ssim_hip_Context* ctx, void* devicePtr, const void* hostPtr, size_t size) RMGR_NOEXCEPT;
rmgr_int32_t rmgr_ssim_hip_memcpy_d2h(rmgr_ssim_hip_Context* ctx, void* hostPtr, const void* devicePtr, size_t size) RMGR_NOEXCEPT;

/*
 * Kernel timing with HIP events on the context's stream.  While enabled, every launch of the
 * main SSIM kernel is bracketed by an event pair; get_profile() (after a synchronize) returns
 * how many launches were timed and their summed duration in milliseconds, then resets.
 * "Launch" means kernel launch, not API call: a host-pointer call that asks for a map is pipelined
 * over row bands (one kernel launch per band) and therefore counts as several launches.
 */
rmgr_int32_t rmgr_ssim_hip_set_profiling(rmgr_ssim_hip_Context* ctx, rmgr_int32_t enabled) RMGR_NOEXCEPT;
rmgr_int32_t rmgr_ssim_hip_get_profile(rmgr_ssim_hip_Context* ctx, rmgr_uint64_t* launches, double* kernelMs) RMGR_NOEXCEPT;
/* The shader clock the profiled launches really ran at (round 6).  While profiling is enabled, the first workgroups of every strip-kernel launch -- one per XCD: the
 * dispatcher deals consecutive workgroups to the XCDs round robin -- read s_memtime (one tick per shader cycle) and s_memrealtime (the constant reference clock,
 * hipDeviceAttributeWallClockRate: 100 MHz) when they start and when they end and add the differences to per-XCD device counters; this call (it waits for the stream)
 * returns *shaderMHz = the mean over the XCDs of reference rate x cycles / ticks, *slowestXcdMHz = the lowest XCD's (a launch ends when its slowest XCD does) and how many
 * launches contributed (0 and 0.0 if none), then clears the counters.  Boxes of one pool differ in the clock they sustain under THIS kernel's load by more than they
 * differ under a pure FMA stream; lane-operations per CYCLE are what a kernel change changes.  Any pointer may be NULL. */
rmgr_int32_t rmgr_ssim_hip_get_profile_clock(rmgr_ssim_hip_Context* ctx, double* shaderMHz, double* slowestXcdMHz, rmgr_uint64_t* launches) RMGR_NOEXCEPT;

/*
 * What the vector ALUs of the context's device sustain RIGHT NOW at a forced occupancy (profiling aid; no reference counterpart; not on
 * the SSIM path).  A pure packed-fp32 instruction stream runs with its register footprint padded so that the hardware cannot place more
 * than wavesPerSimd (1, 2, 3, 4 or 8) wavefronts on a SIMD, on a grid of exactly the device's capacity at that occupancy; streamKind 0:
 * independent v_pk_fma_f32 (the issue peak at that occupancy), 1: two interleaved dependent chains of six (the blur's row sums).  Twenty untimed
 * launches (~40 ms: the clock leaves its idle state), then `launches` (1 ... 64) timed ones of about 2 ms each between HIP events on the context's stream, all
 * enqueued back to back and waited for once (a host-side wait between launches is an idle gap after which the clock ramps again); three such bursts, the best
 * one counts (about one burst in four runs in a degraded mode -- same clock, the rate of one wave fewer per SIMD: profiles/r06_probe_bimodal.txt); *teraLaneOps
 * receives the best burst's MEDIAN launch's rate in 10^12 lane-operations per second (128 per packed instruction and wavefront).  The strip kernels are
 * fp32-VALU bound and run at two (modes 0, 1, 3) or three (modes 2, 4) wavefronts per SIMD: bench.py divides their lane-operations per
 * second by this figure, measured in the same process right before and right after the timed steps, instead of by a constant measured on
 * another box.  shaderMHz / slowestXcdMHz (may be NULL): the shader clock the timed launches really ran at -- mean over the XCDs, and the slowest XCD's --
 * measured on the device as rmgr_ssim_hip_get_profile_clock measures the strip kernels'.  Blocking.  EINVAL for any other occupancy or stream kind.
 */
rmgr_int32_t rmgr_ssim_hip_probe_valu(rmgr_ssim_hip_Context* ctx, rmgr_int32_t wavesPerSimd, rmgr_int32_t streamKind, rmgr_int32_t launches,
                                      double* teraLaneOps, double* shaderMHz, double* slowestXcdMHz) RMGR_NOEXCEPT;

/*
 * The synthetic test pattern the benchmark and the self-tests run on (no reference counterpart: the reference's
 * tests read image files), generated in device memory, asynchronously on the context's stream:
 *   r = splitmix64(seed ^ ((y << 32) | x));  g = ((3x + 5y) >> 2) & 255;  A = (3g + (r & 255)) >> 2;
 *   B = clamp(A + ((r >> 8) % 33) - 16, 0, 255)          planar, one byte per pixel, rows stride bytes apart.
 * Known answers: seed 0x5EED gives A[0][0..3] = 45, 59, 51, 6 and B[0][0..3] = 51, 58, 48, 5.
 */
rmgr_int32_t rmgr_ssim_hip_synth_pair_device(rmgr_ssim_hip_Context* ctx, rmgr_uint8_t* imgA, ptrdiff_t strideA, rmgr_uint8_t* imgB, ptrdiff_t strideB,
                                             rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint64_t seed) RMGR_NOEXCEPT;

/* Human-readable description of the device and build ("gfx950 ... CUs ..."); static storage. */
const char* rmgr_ssim_hip_describe(rmgr_ssim_hip_Context* ctx) RMGR_NOEXCEPT;

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* RMGR_SSIM_HIP_H */
