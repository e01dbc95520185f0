// ssim_hip_abi.cpp -- implementation of the C ABI declared in include/rmgr/ssim-hip.h.
//
// Host-side driver of the GPU path: what src/ssim.cpp:933-1106 (compute_ssim) is to the
// reference's tile kernels, this file is to ssim_kernels.hip -- parameter validation with the
// reference's error codes, staging, launch, final mean.  No CPU arithmetic fallback exists: when
// no gfx950 device is usable every entry point fails loudly with ENODEV.
#include <rmgr/ssim-hip.h>
#include "ssim_kernels.h"
#include <rccl/rccl.h>      // types only: the library is dlopen()ed on first use, never linked
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <deque>
#include <string>
#include <vector>

using ssim_hip::PairDesc;

struct rmgr_ssim_hip_Context_ {
    int         device;
    int         cu_count;
    int         xcd_count;      // hipDeviceAttributeNumberOfXccs (8 on MI355X): the modulus of the kernels' XCD-aware workgroup order
    hipStream_t stream;
    bool        owns_stream;
    int         mode;
    int         strip_rows;
    int         variant;

    // grow-only device scratch
    double*   partials;     size_t partials_cap;   // doubles
    // Batch descriptor tables: a small ring of (device table, pinned mirror, "last launch that read it" event), so
    // that enqueueing a DIFFERENT batch never waits for the stream -- only for the launch kDescSlots enqueues ago --
    // and a serving loop that alternates between a few batches re-uses their uploaded tables.
    enum { kDescSlots = 4 };
    struct DescSlot {
        PairDesc* dev;  size_t dev_cap;     // entries
        PairDesc* host; size_t host_cap;    // pinned mirror of what `dev` holds (or will hold once the queued copy ran)
        size_t    live;                     // entries of `host`/`dev` that are valid (0: nothing uploaded)
        hipEvent_t used;                    // recorded after the last launch reading `dev`
        bool      in_flight;
    } desc_slots[kDescSlots];
    int       desc_next;
    uint8_t*  stage_a;      size_t stage_a_cap;    // bytes (host-pointer path)
    uint8_t*  stage_b;      size_t stage_b_cap;
    float*    stage_map;    size_t stage_map_cap;  // floats
    // pinned host scratch
    double*   h_sums;       size_t h_sums_cap;     // doubles: per-image sums of the blocking entry points, written by the GPU
    // pipelined host batches: two device slots, two pinned gather buffers, a copy stream and per-slot events
    uint8_t*  slot_dev[2];  size_t slot_dev_cap[2];
    uint8_t*  slot_pin[2];  size_t slot_pin_cap[2];
    double*   batch_sums;   size_t batch_sums_cap;
    hipStream_t copy_stream;
    hipEvent_t  slot_copied[2], slot_done[2];
    // banded single-pair host calls: a third stream for the map on its way back, per-band events
    enum { kMaxBands = 16 };
    hipStream_t out_stream;
    hipEvent_t  band_copied[kMaxBands], band_done[kMaxBands];
    uint8_t*  h_stage;      size_t h_stage_cap;    // bytes: small image pairs are gathered here for one DMA
    float*    h_map[2];     size_t h_map_cap[2];   // floats: bounce buffers for the map copy-back
    hipEvent_t map_ev[2];

    bool profiling;
    uint64_t* clock_dev;      // kClockWords device counters the profiled strip launches' first workgroups (one per XCD) add their shader cycles / reference ticks to (ssim_kernels.hip clock_begin); NULL until profiling is first enabled
    int       wall_clock_khz; // rate of s_memrealtime (hipDeviceAttributeWallClockRate; 100 MHz on MI355X)
    std::vector<std::pair<hipEvent_t, hipEvent_t> > pending;   // recorded, not yet read
    std::vector<std::pair<hipEvent_t, hipEvent_t> > free_events;
    uint64_t prof_launches;
    double   prof_ms;

    ncclComm_t comm;          // RCCL communicator (rmgr_ssim_hip_comm_*), NULL until comm_init
    // every queued all-reduce is bracketed by two events on the stream: what is still outstanding, oldest first.  The deadline of
    // rmgr_ssim_hip_synchronize / _destroy applies to ONE collective from the moment its turn has come (its `begin` event is complete) --
    // never to the kernels queued around it (30 s of legitimate work used to get a healthy communicator aborted)
    struct Collective { hipEvent_t begin, end; bool started; std::chrono::steady_clock::time_point since; };
    std::deque<Collective> collectives;
    std::vector<hipEvent_t> spare_events;
    bool       comm_nonblocking;   // created with ncclCommInitRankConfig(blocking = 0): calls may report "in progress"
    int        comm_ranks;         // ncclCommCount of the communicator

    // rmgr_ssim_hip_tune: the measured choice per launch shape, consulted by enqueue() while the context is on its default tuning
    struct Tuned { uint32_t width, height, count; int mode; bool map; int variant, strip_rows; };
    std::vector<Tuned> tuned;

    char describe[256];
    std::mutex lock;
};

namespace {

const size_t kSmallStageBytes = size_t(768) << 10;   // image pairs up to this many bytes go through one pinned gather copy

int map_hip_error(hipError_t e)
{
    switch (e) {
    case hipSuccess:                return 0;
    case hipErrorOutOfMemory:       return ENOMEM;
    case hipErrorNoDevice:
    case hipErrorInvalidDevice:
    case hipErrorInsufficientDriver:
    case hipErrorNoBinaryForGpu:
    case hipErrorInvalidDeviceFunction:
        return ENODEV;
    case hipErrorInvalidValue:      return EINVAL;
    default:                        return ECHILD;   // "an error occurred in a worker" (src/ssim.cpp:1096-1097)
    }
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); return map_hip_error(e_); } } while (0)

// Entry points run on the context's device WITHOUT changing the calling thread's current device (the reference API
// has no such side effect): the previous device is restored on every exit path.
struct DeviceGuard {
    int prev, rc;
    bool restore;
    DeviceGuard() : prev(-1), rc(0), restore(false) {}          // inactive until enter()
    explicit DeviceGuard(int dev) : prev(-1), rc(0), restore(false) { enter(dev); }
    int enter(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != dev) {
            const hipError_t e = hipSetDevice(dev);
            if (e != hipSuccess) { (void)hipGetLastError(); rc = map_hip_error(e); return rc; }
            restore = prev >= 0;
        }
        return 0;
    }
    ~DeviceGuard() { if (restore) (void)hipSetDevice(prev); }
private:
    DeviceGuard(const DeviceGuard&);
    DeviceGuard& operator=(const DeviceGuard&);
};
#define USE_DEVICE(c) DeviceGuard device_guard_((c)->device); if (device_guard_.rc) return device_guard_.rc

template <typename T>
int grow_device(T*& ptr, size_t& cap, size_t need)
{
    if (need <= cap) return 0;
    if (ptr) { HIP_TRY(hipFree(ptr)); ptr = NULL; cap = 0; }
    size_t n = need + need / 4 + 64;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T)));
    cap = n;
    return 0;
}

template <typename T>
int grow_pinned(T*& ptr, size_t& cap, size_t need)
{
    if (need <= cap) return 0;
    if (ptr) { HIP_TRY(hipHostFree(ptr)); ptr = NULL; cap = 0; }
    size_t n = need + need / 4 + 64;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T), hipHostMallocDefault));
    cap = n;
    return 0;
}

// The reference's parameter checks, in its order (src/ssim.cpp:962-978).
int validate(const float* ssim, const rmgr_ssim_Params* p, const rmgr_ssim_ThreadPool* tp)
{
    if (p == NULL) return EINVAL;                                    // src/ssim.cpp:1147-1151
    if (ssim == NULL && p->ssimMap == NULL) return EINVAL;
    if (p->imgA.topLeft == NULL || p->imgB.topLeft == NULL) return EINVAL;
    if (tp != NULL && tp->dispatch != NULL && tp->threadCount == 0u) return EINVAL;
    return 0;
}

// Byte extent [lo, hi] (inclusive, relative to topLeft) touched by a width x height image.
void extent(const rmgr_ssim_ImgParams& im, uint32_t w, uint32_t h, int64_t& lo, int64_t& hi)
{
    const int64_t dx = (int64_t)(w - 1) * (int64_t)im.step, dy = (int64_t)(h - 1) * (int64_t)im.stride;
    lo = (dx < 0 ? dx : 0) + (dy < 0 ? dy : 0);
    hi = (dx > 0 ? dx : 0) + (dy > 0 ? dy : 0);
}

// Event pair for timing one launch (NULLs when profiling is off).  The pair enters `pending` only after BOTH events
// were recorded (commit_events); a launch that fails, or launches nothing, hands it back (release_events).
int acquire_events(rmgr_ssim_hip_Context* c, hipEvent_t& b, hipEvent_t& e)
{
    b = e = NULL;
    if (!c->profiling) return 0;
    if (!c->free_events.empty()) {
        b = c->free_events.back().first; e = c->free_events.back().second;
        c->free_events.pop_back();
        return 0;
    }
    HIP_TRY(hipEventCreate(&b));
    hipError_t err = hipEventCreate(&e);
    if (err != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(b); b = e = NULL; return map_hip_error(err); }
    return 0;
}

void release_events(rmgr_ssim_hip_Context* c, hipEvent_t b, hipEvent_t e)
{
    if (!b) return;
    try { c->free_events.push_back(std::make_pair(b, e)); }
    catch (...) { (void)hipEventDestroy(b); (void)hipEventDestroy(e); }
}

int commit_events(rmgr_ssim_hip_Context* c, hipEvent_t b, hipEvent_t e)
{
    if (!b) return 0;
    try { c->pending.push_back(std::make_pair(b, e)); }
    catch (...) { (void)hipEventDestroy(b); (void)hipEventDestroy(e); return ENOMEM; }
    return 0;
}

// Reads every pending pair.  A pair that cannot be read is dropped (its events recycled) and reported once; the
// pairs before and after it are counted exactly once.
int drain_profile(rmgr_ssim_hip_Context* c)
{
    int rc = 0;
    for (size_t i = 0; i < c->pending.size(); ++i) {
        float ms = 0.f;
        hipError_t err = hipEventSynchronize(c->pending[i].second);
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, c->pending[i].first, c->pending[i].second);
        if (err == hipSuccess) {
            c->prof_ms += ms;
            c->prof_launches += 1;
        } else {
            (void)hipGetLastError();
            if (!rc) rc = map_hip_error(err);
        }
        release_events(c, c->pending[i].first, c->pending[i].second);
    }
    c->pending.clear();
    return rc;
}

// Device copy of a batch's descriptor table: re-used when any ring slot already holds exactly these descriptors,
// otherwise uploaded into the next slot (waiting, at most, for the launch that read that slot kDescSlots batches ago).
int upload_descs(rmgr_ssim_hip_Context* c, const PairDesc* descs, uint32_t count, int& slot_out)
{
    typedef rmgr_ssim_hip_Context_::DescSlot Slot;
    for (int i = 0; i < rmgr_ssim_hip_Context_::kDescSlots; ++i) {
        Slot& s = c->desc_slots[i];
        if (s.live == count && memcmp(s.host, descs, sizeof(PairDesc) * count) == 0) { slot_out = i; return 0; }
    }
    const int k = c->desc_next;
    c->desc_next = (k + 1) % rmgr_ssim_hip_Context_::kDescSlots;
    Slot& s = c->desc_slots[k];
    if (s.in_flight) { HIP_TRY(hipEventSynchronize(s.used)); s.in_flight = false; }
    s.live = 0;
    int rc;
    if ((rc = grow_device(s.dev, s.dev_cap, count))) return rc;
    if ((rc = grow_pinned(s.host, s.host_cap, count))) return rc;
    if (!s.used) HIP_TRY(hipEventCreateWithFlags(&s.used, hipEventDisableTiming));
    memcpy(s.host, descs, sizeof(PairDesc) * count);
    HIP_TRY(hipMemcpyAsync(s.dev, s.host, sizeof(PairDesc) * count, hipMemcpyHostToDevice, c->stream));
    s.live = count;
    slot_out = k;
    return 0;
}

// Enqueue kernel + reduction for `count` pairs whose descriptors are in `descs` (host).
// y_begin / y_rows / reduce: the row window of this launch (ssim_kernels.h plan()); the default is the whole image.
// cells_out (row-band entry point only): the strips write their cell partials there instead of into the context's scratch.
int enqueue(rmgr_ssim_hip_Context* c, uint32_t width, uint32_t height, uint32_t count, const PairDesc* descs, bool any_map, double* sums_dev,
            uint32_t y_begin = 0, uint32_t y_rows = 0xFFFFFFFFu, bool reduce = true, double* cells_out = NULL)
{
    int variant = c->variant, strip_rows = c->strip_rows;
    if (variant == 0 && strip_rows == 0 && y_begin == 0 && y_rows == 0xFFFFFFFFu && !cells_out) {      // default tuning, whole images: a measured choice, if one was made
        for (size_t i = 0; i < c->tuned.size(); ++i) {
            const rmgr_ssim_hip_Context_::Tuned& t = c->tuned[i];
            if (t.width == width && t.height == height && t.count == count && t.mode == c->mode && t.map == any_map) { variant = t.variant; strip_rows = t.strip_rows; break; }
        }
    }
    if (variant == 0 && strip_rows == 0) variant = ssim_hip::default_variant(width, height, count, c->mode, c->cu_count);
    bool all_fit = true;
    for (uint32_t i = 0; i < count && all_fit; ++i)
        all_fit = ssim_hip::fits_strip2(descs[i], width, height);
    if (!all_fit) variant = 1;
    ssim_hip::Geometry geo = ssim_hip::plan(width, height, count, c->mode, strip_rows, variant, c->cu_count, c->xcd_count, y_begin, y_rows);
    geo.wide = !all_fit;                             // the 64-bit form of the one-column kernel only where it is needed
    geo.map_unit = any_map && (width & 1u) == 0;     // the 8-byte map stores of the two-column kernel (ssim_kernels.hip, MAP == 2)
    for (uint32_t i = 0; i < count && geo.map_unit; ++i)
        geo.map_unit = descs[i].map != NULL && descs[i].map_step == 1;
    int rc = cells_out ? 0 : grow_device(c->partials, c->partials_cap, ssim_hip::partials_size(geo));
    if (rc) return rc;
    PairDesc single = descs[0];
    const PairDesc* descs_dev = NULL;
    int slot = -1;
    if (count > 1) {
        if ((rc = upload_descs(c, descs, count, slot))) return rc;
        descs_dev = c->desc_slots[slot].dev;
        if (any_map && !single.map) single.map = reinterpret_cast<float*>(1);  // only its non-NULLness is used
    }
    const bool launches_kernel = count > 0 && geo.strips_x > 0 && geo.strips_y > 0;
    hipEvent_t eb = NULL, ee = NULL;
    if (launches_kernel && (rc = acquire_events(c, eb, ee))) return rc;
    const hipError_t err = ssim_hip::launch(geo, c->mode, variant, ssim_hip::interleaved_group(descs, count), descs_dev, single, cells_out ? cells_out : c->partials, sums_dev, c->stream, eb, ee, reduce, c->profiling ? c->clock_dev : NULL);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        release_events(c, eb, ee);
        if (slot >= 0) { (void)hipStreamSynchronize(c->stream); c->desc_slots[slot].live = 0; }   // the table upload may still be queued
        return map_hip_error(err);
    }
    // the launch is queued and reads the slot's device table: mark the slot busy BEFORE anything else can fail
    if (slot >= 0) {
        const hipError_t e = hipEventRecord(c->desc_slots[slot].used, c->stream);
        if (e != hipSuccess) {              // cannot track the reader: wait for it instead of leaving the table unprotected
            (void)hipGetLastError();
            (void)hipStreamSynchronize(c->stream);
            release_events(c, eb, ee);
            return map_hip_error(e);
        }
        c->desc_slots[slot].in_flight = true;
    }
    if ((rc = commit_events(c, eb, ee))) return rc;
    return 0;
}

PairDesc make_desc(const rmgr_ssim_Params& p)
{
    PairDesc d;
    d.a = p.imgA.topLeft; d.a_step = p.imgA.step; d.a_stride = p.imgA.stride;
    d.b = p.imgB.topLeft; d.b_step = p.imgB.step; d.b_stride = p.imgB.stride;
    d.map = p.ssimMap;
    d.map_step = p.ssimMap ? p.ssimStep : 0;      // src/ssim.cpp:980-987
    d.map_stride = p.ssimMap ? p.ssimStride : 0;
    return d;
}

float mean_of(double sum, uint32_t width, uint32_t height)
{
    return float(sum / double(uint32_t(width * height)));   // src/ssim.cpp:1102 (32-bit product kept)
}

// ---- the per-pixel map on its way back to the caller (host-pointer entry points) ----

// Rows [y0, y1) of the dense device map (c->stage_map, W floats per row) into the caller's map, any ssimStep /
// ssimStride.  Unit step and a positive stride: the DMA engine writes the caller's rows directly (one 2-D copy).
// Anything else: through two pinned bounce buffers in chunks, the CPU scattering chunk k-1 while chunk k is in
// flight.  Returns after the rows are in the caller's memory.
int map_rows_to_host(rmgr_ssim_hip_Context* c, const rmgr_ssim_Params& p, uint32_t y0, uint32_t y1, hipStream_t stream)
{
    const uint32_t W = p.width;
    if (y1 <= y0 || W == 0) return 0;
    const float* src = c->stage_map + (size_t)y0 * W;
    if (p.ssimStep == 1 && p.ssimStride >= (ptrdiff_t)W) {
        float* dst = p.ssimMap + (ptrdiff_t)y0 * p.ssimStride;
        if (p.ssimStride == (ptrdiff_t)W)
            HIP_TRY(hipMemcpyAsync(dst, src, sizeof(float) * (size_t)(y1 - y0) * W, hipMemcpyDeviceToHost, stream));
        else
            HIP_TRY(hipMemcpy2DAsync(dst, sizeof(float) * (size_t)p.ssimStride, src, sizeof(float) * W, sizeof(float) * W, y1 - y0, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return 0;
    }
    int rc;
    const size_t rowsPerChunk = std::max<size_t>(1, (size_t(8) << 20) / (sizeof(float) * W));
    const size_t chunkFloats = rowsPerChunk * W;
    if ((rc = grow_pinned(c->h_map[0], c->h_map_cap[0], chunkFloats))) return rc;
    if ((rc = grow_pinned(c->h_map[1], c->h_map_cap[1], chunkFloats))) return rc;
    for (int i = 0; i < 2; ++i)
        if (!c->map_ev[i]) HIP_TRY(hipEventCreateWithFlags(&c->map_ev[i], hipEventDisableTiming));
    const size_t total = y1 - y0, chunks = (total + rowsPerChunk - 1) / rowsPerChunk;
    for (size_t k = 0; k <= chunks; ++k) {
        if (k < chunks) {
            const size_t r0 = k * rowsPerChunk, rows = std::min(rowsPerChunk, total - r0);
            HIP_TRY(hipMemcpyAsync(c->h_map[k & 1], src + r0 * W, sizeof(float) * rows * W, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipEventRecord(c->map_ev[k & 1], stream));
        }
        if (k > 0) {
            const size_t j = k - 1, r0 = j * rowsPerChunk, rows = std::min(rowsPerChunk, total - r0);
            HIP_TRY(hipEventSynchronize(c->map_ev[j & 1]));
            const float* from = c->h_map[j & 1];
            for (size_t y = 0; y < rows; ++y, from += W) {
                float* row = p.ssimMap + (ptrdiff_t)(y0 + r0 + y) * p.ssimStride;
                if (p.ssimStep == 1) memcpy(row, from, sizeof(float) * W);
                else for (uint32_t x = 0; x < W; ++x) row[(ptrdiff_t)x * p.ssimStep] = from[x];
            }
        }
    }
    return 0;
}

// Rows of both images occupy disjoint byte ranges (no column-major or otherwise row-interleaved layout), and the
// image is tall enough to be worth cutting: the precondition of the banded pipeline.
bool bandable(const rmgr_ssim_Params& p)
{
    if (p.height < 256 || p.width == 0) return false;
    const rmgr_ssim_ImgParams* im[2] = {&p.imgA, &p.imgB};
    for (int i = 0; i < 2; ++i) {
        const int64_t span = (int64_t)(p.width - 1) * (im[i]->step < 0 ? -(int64_t)im[i]->step : (int64_t)im[i]->step) + 1;
        const int64_t stride = im[i]->stride < 0 ? -(int64_t)im[i]->stride : (int64_t)im[i]->stride;
        if (stride < span) return false;
    }
    return true;
}

// Byte range [lo, hi] (relative to topLeft) of rows [r0, r1) of a row-separable image.
void rows_extent(const rmgr_ssim_ImgParams& im, uint32_t w, uint32_t r0, uint32_t r1, int64_t& lo, int64_t& hi)
{
    const int64_t dx = (int64_t)(w - 1) * (int64_t)im.step;
    const int64_t ya = (int64_t)r0 * (int64_t)im.stride, yb = (int64_t)(r1 - 1) * (int64_t)im.stride;
    lo = std::min(ya, yb) + (dx < 0 ? dx : 0);
    hi = std::max(ya, yb) + (dx > 0 ? dx : 0);
}

// The banded pipeline of one large host pair (see the call site), with or without a map.  `dev`/`d` address the staged
// device copies (not yet filled), loA/loB are the byte offsets of the images' lowest addresses relative to topLeft.
// Without a map there is nothing to send back and no helper thread: the bands only hide the kernel behind the H2D copy
// of the following band.  Measured (round 3, profiles/r03_host_probe.txt, on a build that banded every map-less call): every
// additional pageable copy costs ~20 us, more than the ~12 us of kernel a band hides at 4096^2 (0.742 ms unbanded, 0.762 /
// 0.790 / 0.891 ms at 2 / 4 / 8 bands); it pays from 8192^2 on (2.795 -> 2.650 ms at 4 bands), so by default only such images
// take this path without a map ($RMGR_SSIM_HIP_BANDS forces it for any size: tools/host_call_probe.py).
int compute_banded(rmgr_ssim_hip_Context* c, const rmgr_ssim_Params& p, const rmgr_ssim_Params& dev, const PairDesc& d, int64_t loA, int64_t loB)
{
    (void)dev;
    const uint32_t W = p.width, H = p.height;
    // Band size: ~2 Mpixel (8 MB of map) per band, at most kMaxBands -- measured on MI355X / PCIe Gen5 (profiles/
    // r02_host_probe.txt): 4096^2 10.6 Gpix/s at 6-8 bands (8.2 unpipelined), 8192^2 12.0 at 16 (8.7), 2048^2 8.1 at 2 (7.0).
    int bands = (int)std::min<uint64_t>(((uint64_t)W * H + (1u << 21) - 1) >> 21, rmgr_ssim_hip_Context_::kMaxBands);
    if (!p.ssimMap) bands = std::min(bands, 4);
    if (const char* e = getenv("RMGR_SSIM_HIP_BANDS")) bands = atoi(e);
    bands = std::max(1, std::min<int>(bands, rmgr_ssim_hip_Context_::kMaxBands));
    const uint32_t cell = ssim_hip::cell_rows_for(H);         // windows start on reduction-cell boundaries
    uint32_t band_rows = ((H + bands - 1) / bands + cell - 1) & ~(cell - 1);
    if (band_rows < 64) band_rows = 64;
    const int n = (int)((H + band_rows - 1) / band_rows);
    if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->out_stream) HIP_TRY(hipStreamCreateWithFlags(&c->out_stream, hipStreamNonBlocking));
    for (int k = 0; k < n; ++k) {
        if (!c->band_copied[k]) HIP_TRY(hipEventCreateWithFlags(&c->band_copied[k], hipEventDisableTiming));
        if (!c->band_done[k]) HIP_TRY(hipEventCreateWithFlags(&c->band_done[k], hipEventDisableTiming));
    }
    // input chunk k = rows [in[k], in[k+1]); once it has landed, output rows [out[k], out[k+1]) can be computed:
    // everything up to one reduction cell (>= the 5 halo rows) short of the rows present.
    uint32_t in[rmgr_ssim_hip_Context_::kMaxBands + 1], out[rmgr_ssim_hip_Context_::kMaxBands + 1];
    for (int k = 0; k <= n; ++k) {
        in[k] = std::min<uint64_t>((uint64_t)k * band_rows, H);
        out[k] = (k == 0) ? 0 : (k == n ? H : in[k] - cell);
    }

    // The helper thread returns band k's map rows while the caller's thread feeds bands k+1, k+2, ...
    struct Shared {
        std::mutex m; std::condition_variable cv;
        int launched; bool abort; int rc;
    } sh;
    sh.launched = 0; sh.abort = false; sh.rc = 0;
    struct Worker {
        rmgr_ssim_hip_Context* c; const rmgr_ssim_Params* p; Shared* sh; const uint32_t* out; int n;
        void operator()() const
        {
            int rc = 0;
            if (hipSetDevice(c->device) != hipSuccess) { (void)hipGetLastError(); rc = ENODEV; }
            for (int k = 0; k < n && rc == 0; ++k) {
                {
                    std::unique_lock<std::mutex> lk(sh->m);
                    sh->cv.wait(lk, [&] { return sh->launched > k || sh->abort; });
                    if (sh->launched <= k) break;            // aborted before band k was launched
                }
                const hipError_t e = hipEventSynchronize(c->band_done[k]);
                if (e != hipSuccess) { (void)hipGetLastError(); rc = map_hip_error(e); break; }
                rc = map_rows_to_host(c, *p, out[k], out[k + 1], c->out_stream);
            }
            std::lock_guard<std::mutex> lk(sh->m);
            if (rc && !sh->rc) sh->rc = rc;
        }
    };
    const Worker work = {c, &p, &sh, out, n};
    const bool with_map = p.ssimMap != NULL;
    std::thread helper;
    bool threaded = with_map;
    if (with_map) { try { helper = std::thread(work); } catch (...) { threaded = false; } }

    int rc = 0;
    const rmgr_ssim_ImgParams* img[2] = {&p.imgA, &p.imgB};
    uint8_t* stage[2] = {c->stage_a, c->stage_b};
    const int64_t lo_img[2] = {loA, loB};
    for (int k = 0; k < n && rc == 0; ++k) {
        for (int j = 0; j < 2 && rc == 0; ++j) {
            int64_t lo, hi;
            rows_extent(*img[j], W, in[k], in[k + 1], lo, hi);
            const hipError_t e = hipMemcpyAsync(stage[j] + (lo - lo_img[j]), img[j]->topLeft + lo, (size_t)(hi - lo + 1), hipMemcpyHostToDevice, c->copy_stream);
            if (e != hipSuccess) { (void)hipGetLastError(); rc = map_hip_error(e); }
        }
        hipError_t e = rc ? hipSuccess : hipEventRecord(c->band_copied[k], c->copy_stream);
        if (!rc && e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->band_copied[k], 0);
        if (!rc && e != hipSuccess) { (void)hipGetLastError(); rc = map_hip_error(e); }
        if (!rc && out[k + 1] > out[k])
            rc = enqueue(c, W, H, 1, &d, with_map, c->h_sums, out[k], out[k + 1] - out[k], k == n - 1);
        if (!rc) {
            e = hipEventRecord(c->band_done[k], c->stream);
            if (e != hipSuccess) { (void)hipGetLastError(); rc = map_hip_error(e); }
        }
        if (!rc) {
            { std::lock_guard<std::mutex> lk(sh.m); sh.launched = k + 1; }
            sh.cv.notify_one();
        }
    }
    if (rc) {
        { std::lock_guard<std::mutex> lk(sh.m); sh.abort = true; }
        sh.cv.notify_one();
    }
    if (threaded) helper.join();
    else if (!rc && with_map) work();             // no thread could be started: the same steps, serially
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipGetLastError(); if (!rc) rc = map_hip_error(e); }
    (void)hipStreamSynchronize(c->copy_stream);
    return rc ? rc : sh.rc;
}

int comm_bounded_sync(rmgr_ssim_hip_Context* c);      // with the RCCL section below
void comm_forget_collectives(rmgr_ssim_hip_Context* c);

// ---- the caller's thread pool (rmgr_ssim_ThreadPool) ---------------------------------------------------------------------------
// The reference cuts the image into 256 x 64 tiles and hands them to threadPool->dispatch as jobs: `fct` is to be called exactly jobCount
// times with jobNum in [0, jobCount), on up to threadCount threads, args[t] used by one thread at a time; a non-zero return becomes ECHILD
// -- when a global value was asked for (src/ssim.cpp:1048-1100, include/rmgr/ssim.h:448-466).  Rounds 1-5 validated the pool and never
// called it.  Here the contract is kept with the unit of work this engine has on the HOST side: a ROW BAND of the pair -- its source rows
// (with the 5-row halo) to the device, the strips of the band (a row window of the launch: cells at absolute positions, so any order and
// any number of threads give the one-launch result bit for bit), and, with a map, the band's map rows back into the caller's buffer.  Jobs
// are independent and may run in any order; the context itself is single-threaded, so a job holds the call's lock while it enqueues and
// lets go of it while it waits for its map rows -- with several pool threads one band's copy-back overlaps the next band's copy-in and
// kernel, as in the library's own pipeline (compute_banded()).  After dispatch returns: the fixed-order reduction of the cells, the mean.
struct PoolCall {
    rmgr_ssim_hip_Context* c;
    const rmgr_ssim_Params* host;      // the caller's parameters (host pointers)
    PairDesc d;                        // the staged pair (device pointers)
    int64_t loA, loB;                  // byte offsets of the images' lowest addresses relative to topLeft
    uint32_t jobs, band_rows;
    bool whole;                        // the layout cannot be cut into row bands: one job stages the whole byte ranges
    std::mutex m;
    int rc;                            // first error of a job (errno)
    uint32_t ran;                      // jobs that ran to their end
    char done[rmgr_ssim_hip_Context_::kMaxBands];
};

void pool_job(void* arg, rmgr_uint32_t jobNum) RMGR_NOEXCEPT
{
    PoolCall& pc = **static_cast<PoolCall**>(arg);
    rmgr_ssim_hip_Context* c = pc.c;
    const rmgr_ssim_Params& p = *pc.host;
    const uint32_t W = p.width, H = p.height;
    int rc = 0;
    DeviceGuard on_device(c->device);            // a pool thread has a current device of its own
    std::unique_lock<std::mutex> lk(pc.m);
    if (jobNum >= pc.jobs || pc.done[jobNum]) { if (!pc.rc) pc.rc = ECHILD; return; }      // a pool that invents or repeats jobs: "an error occurred in a worker"
    if (on_device.rc) rc = on_device.rc;
    const uint32_t y0 = jobNum * pc.band_rows, y1 = (uint32_t)std::min<uint64_t>((uint64_t)y0 + pc.band_rows, H);
    if (!rc) {
        const rmgr_ssim_ImgParams* img[2] = {&p.imgA, &p.imgB};
        uint8_t* stage[2] = {const_cast<uint8_t*>(pc.d.a) + pc.loA, const_cast<uint8_t*>(pc.d.b) + pc.loB};      // = the staging buffers' first bytes
        const int64_t lo_img[2] = {pc.loA, pc.loB};
        for (int j = 0; j < 2 && !rc; ++j) {
            int64_t lo, hi;
            if (pc.whole) extent(*img[j], W, H, lo, hi);
            else rows_extent(*img[j], W, y0 >= 5 ? y0 - 5 : 0, (uint32_t)std::min<uint64_t>((uint64_t)y1 + 5, H), lo, hi);      // the band's rows and its 5-row halo
            const hipError_t e = hipMemcpyAsync(stage[j] + (lo - lo_img[j]), img[j]->topLeft + lo, (size_t)(hi - lo + 1), hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess) { (void)hipGetLastError(); rc = map_hip_error(e); }
        }
    }
    if (!rc) rc = enqueue(c, W, H, 1, &pc.d, p.ssimMap != NULL, c->h_sums, y0, y1 - y0, false);
    if (!rc && p.ssimMap) {
        const bool direct = p.ssimStep == 1 && p.ssimStride >= (ptrdiff_t)W;       // the DMA engine writes the caller's rows itself: no shared bounce buffers
        if (direct) {
            hipError_t e = c->band_done[jobNum] ? hipSuccess : hipEventCreateWithFlags(&c->band_done[jobNum], hipEventDisableTiming);
            if (e == hipSuccess && !c->out_stream) e = hipStreamCreateWithFlags(&c->out_stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventRecord(c->band_done[jobNum], c->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->out_stream, c->band_done[jobNum], 0);
            if (e != hipSuccess) { (void)hipGetLastError(); rc = map_hip_error(e); }
            if (!rc) {
                // the copy is queued behind this band's kernel on the third stream; the wait happens WITHOUT the call's lock
                float* dst = p.ssimMap + (ptrdiff_t)y0 * p.ssimStride;
                const float* src = c->stage_map + (size_t)y0 * W;
                if (p.ssimStride == (ptrdiff_t)W) e = hipMemcpyAsync(dst, src, sizeof(float) * (size_t)(y1 - y0) * W, hipMemcpyDeviceToHost, c->out_stream);
                else e = hipMemcpy2DAsync(dst, sizeof(float) * (size_t)p.ssimStride, src, sizeof(float) * W, sizeof(float) * W, y1 - y0, hipMemcpyDeviceToHost, c->out_stream);
                hipEvent_t mine = NULL;
                if (e == hipSuccess) e = hipEventCreateWithFlags(&mine, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventRecord(mine, c->out_stream);
                lk.unlock();
                if (e == hipSuccess) e = hipEventSynchronize(mine);
                if (mine) (void)hipEventDestroy(mine);
                if (e != hipSuccess) { (void)hipGetLastError(); rc = map_hip_error(e); }
                lk.lock();
            }
        } else {
            rc = map_rows_to_host(c, p, y0, y1, c->stream);      // through the context's two bounce buffers: under the lock
        }
    }
    if (rc) { if (!pc.rc) pc.rc = rc; return; }
    pc.done[jobNum] = 1;
    ++pc.ran;
}

// compute_ssim on host pointers through the caller's pool.  `dev` / `d`: the staged pair (buffers grown, nothing copied yet).
int compute_via_pool(rmgr_ssim_hip_Context* c, float* ssim, const rmgr_ssim_Params& p, const PairDesc& d, int64_t loA, int64_t loB, const rmgr_ssim_ThreadPool& tp)
{
    const uint32_t W = p.width, H = p.height;
    const uint32_t maxThreadCount = sizeof(void*) * 8;                               // the reference's cap (src/ssim.cpp:1025)
    const uint32_t threads = std::min<uint32_t>(tp.threadCount, maxThreadCount);
    PoolCall pc;
    pc.c = c; pc.host = &p; pc.d = d; pc.loA = loA; pc.loB = loB; pc.rc = 0; pc.ran = 0;
    memset(pc.done, 0, sizeof(pc.done));
    pc.whole = !bandable(p);
    int bands = 1;
    if (W && H && !pc.whole) {
        bands = (int)std::min<uint64_t>(((uint64_t)W * H + (1u << 21) - 1) >> 21, rmgr_ssim_hip_Context_::kMaxBands);       // ~2 Mpixel per band, as compute_banded()
        if (!p.ssimMap) bands = std::min(bands, 4);
        if (const char* e = getenv("RMGR_SSIM_HIP_BANDS")) bands = atoi(e);
        bands = std::max(1, std::min<int>(bands, rmgr_ssim_hip_Context_::kMaxBands));
    }
    const uint32_t cell = ssim_hip::cell_rows_for(H);
    pc.band_rows = std::max<uint32_t>(((H + bands - 1) / bands + cell - 1) & ~(cell - 1), cell);
    pc.jobs = (W && H) ? (H + pc.band_rows - 1) / pc.band_rows : 0;                 // an empty image has no jobs (the reference dispatches its 0 tiles)
    PoolCall* self = &pc;
    void* args[sizeof(void*) * 8];
    for (uint32_t t = 0; t < threads; ++t) args[t] = &self;
    const int poolResult = tp.dispatch(tp.context, pool_job, args, threads, pc.jobs);
    // whatever the pool did, nothing of this call may still be queued when the staging buffers are handed on
    hipError_t e = hipStreamSynchronize(c->stream);
    if (c->out_stream && hipStreamSynchronize(c->out_stream) != hipSuccess) (void)hipGetLastError();
    if (e != hipSuccess) { (void)hipGetLastError(); return map_hip_error(e); }
    if (pc.rc && pc.rc != ECHILD) return pc.rc;                                      // a HIP call failed inside a job
    if (ssim == NULL) return 0;                                                      // src/ssim.cpp:1091: the pool's result is only looked at for the global value
    if (poolResult != 0 || pc.rc || pc.ran != pc.jobs) return ECHILD;                // src/ssim.cpp:1094-1097 (and a pool that ran fewer jobs than it was given)
    double sum = 0.0;
    if (pc.jobs) {
        const ssim_hip::Geometry geo = ssim_hip::plan(W, H, 1, c->mode, c->strip_rows, 0, c->cu_count, c->xcd_count);
        e = ssim_hip::launch_reduce(geo, c->partials, c->partials + geo.partials_per_image(), c->h_sums, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { (void)hipGetLastError(); return map_hip_error(e); }
        sum = c->h_sums[0];
    }
    *ssim = mean_of(sum, W, H);
    return 0;
}

// ---- the process-wide default contexts of the drop-in entry points (ctx == NULL) -------------------------------------
// The reference's compute_ssim() is re-entrant and has no global state (src/ssim.cpp:933-1106): six caller threads get six
// computations running side by side.  Rounds 1-4 ran every ctx == NULL call on ONE default context under its lock: six
// serialised copy-in -> kernel -> copy-out sequences.  Now the default is a small POOL of contexts on the default device
// ($RMGR_SSIM_HIP_DEVICE), each with its own stream, staging buffers and pinned memory: a call leases one for its duration,
// so that one caller's copy-in runs under another's kernel and a third's map on its way back.  Contexts are created on
// demand -- a single-threaded process only ever has one -- up to $RMGR_SSIM_HIP_POOL (default 4, 1 = the old behaviour);
// callers beyond that wait for a lease.  The arithmetic mode of the drop-in calls (rmgr_ssim_hip_set_mode(NULL, ...), i.e.
// rmgr::ssim::select_impl, $RMGR_SSIM_HIP_MODE, the double build's default) is a property of the pool, applied at lease time.
struct DefaultPool {
    std::mutex m;
    std::condition_variable freed;
    std::vector<rmgr_ssim_hip_Context*> all, idle;
    int device, limit, mode, create_err;
    bool configured;
    // memory policy (round 6): staging a context may keep between calls, in bytes (device + pinned); what each context held when its last lease ended
    uint64_t retain_cap;
    struct Held { rmgr_ssim_hip_Context* c; uint64_t device_bytes, pinned_bytes; };
    std::vector<Held> held;
    DefaultPool() : device(0), limit(4), mode(RMGR_SSIM_HIP_MODE_EXACT), create_err(0), configured(false), retain_cap(uint64_t(256) << 20) {}
};
// A LEAKED singleton: neither the pool nor its contexts are ever destroyed.  The process may still be inside a ctx == NULL call when exit() runs
// the static destructors or the library is unloaded (a daemon thread, a detached worker): its Lease must find the mutex, the condition variable
// and the vectors alive when it ends (rounds 1-4 kept a raw pointer for the same reason).
DefaultPool& pool()
{
    static DefaultPool* p = new DefaultPool;
    return *p;
}
#define g_pool (pool())

void pool_configure_locked()
{
    if (g_pool.configured) return;
    g_pool.configured = true;
    if (const char* s = getenv("RMGR_SSIM_HIP_DEVICE")) g_pool.device = atoi(s);
    if (const char* s = getenv("RMGR_SSIM_HIP_POOL")) { const int n = atoi(s); if (n >= 1 && n <= 64) g_pool.limit = n; }
    if (const char* s = getenv("RMGR_SSIM_HIP_POOL_RETAIN_MB")) {       // per context; 0: keep nothing between calls (the reference's behaviour); negative: no cap
        const long long mb = atoll(s);
        g_pool.retain_cap = mb < 0 ? ~uint64_t(0) : (uint64_t)mb << 20;
    }
    const char* m = getenv("RMGR_SSIM_HIP_MODE");
    if (m && atoi(m) >= RMGR_SSIM_HIP_MODE_EXACT && atoi(m) <= RMGR_SSIM_HIP_MODE_SEPARABLE) g_pool.mode = atoi(m);
#if defined(RMGR_SSIM_USE_DOUBLE) && RMGR_SSIM_USE_DOUBLE
    else g_pool.mode = RMGR_SSIM_HIP_MODE_DOUBLE;
#endif
}

// A context of the pool for the duration of one call; blocks while `limit` calls are in flight.
int pool_acquire(rmgr_ssim_hip_Context** out)
{
    std::unique_lock<std::mutex> lk(g_pool.m);
    pool_configure_locked();
    bool create_failed_here = false;     // this call tried to grow the pool and could not (out of memory ...): it waits for a lease instead of trying again
    for (;;) {
        if (!g_pool.idle.empty()) {
            *out = g_pool.idle.back();
            g_pool.idle.pop_back();
            (*out)->mode = g_pool.mode;
            return 0;
        }
        if (g_pool.create_err && g_pool.all.empty()) {
            if (g_pool.create_err == ENODEV) return ENODEV;                         // no usable device: every call fails the same way, at once
            g_pool.create_err = 0;                                                  // anything else (out of memory at that moment ...) is tried again by the next call
        }
        if ((int)g_pool.all.size() < g_pool.limit && !create_failed_here) {
            g_pool.all.push_back(NULL);                                             // reserve a slot; create outside the lock
            lk.unlock();
            rmgr_ssim_hip_Context* c = NULL;
            const int rc = rmgr_ssim_hip_create(&c, g_pool.device, NULL);
            lk.lock();
            if (rc || !c) {
                g_pool.all.erase(std::find(g_pool.all.begin(), g_pool.all.end(), (rmgr_ssim_hip_Context*)NULL));      // one reserved slot (any: they are alike)
                g_pool.create_err = rc ? rc : ENODEV;
                g_pool.freed.notify_all();
                if (g_pool.all.empty()) return g_pool.create_err;
                create_failed_here = true;
                continue;                                                           // others exist (or are being created): wait for one of them
            }
            *std::find(g_pool.all.begin(), g_pool.all.end(), (rmgr_ssim_hip_Context*)NULL) = c;
            c->mode = g_pool.mode;
            *out = c;
            return 0;
        }
        if (g_pool.all.empty()) return g_pool.create_err ? g_pool.create_err : ENODEV;      // nothing exists and nothing is being created: no lease will ever come
        g_pool.freed.wait(lk);                                                      // a context exists or is being created: its release (or its failure) wakes this call
    }
}

// What a context's grow-only staging holds right now (bytes of device memory / of pinned host memory).  Caller owns the context.
void context_held(const rmgr_ssim_hip_Context* c, uint64_t& dev, uint64_t& pin)
{
    dev = (uint64_t)c->partials_cap * sizeof(double) + c->stage_a_cap + c->stage_b_cap + (uint64_t)c->stage_map_cap * sizeof(float)
        + c->slot_dev_cap[0] + c->slot_dev_cap[1] + (uint64_t)c->batch_sums_cap * sizeof(double);
    pin = (uint64_t)c->h_sums_cap * sizeof(double) + c->slot_pin_cap[0] + c->slot_pin_cap[1] + c->h_stage_cap
        + ((uint64_t)c->h_map_cap[0] + c->h_map_cap[1]) * sizeof(float);
    for (int i = 0; i < rmgr_ssim_hip_Context_::kDescSlots; ++i) {
        dev += (uint64_t)c->desc_slots[i].dev_cap * sizeof(PairDesc);
        pin += (uint64_t)c->desc_slots[i].host_cap * sizeof(PairDesc);
    }
}

// Gives a context's grow-only staging back to the system: device scratch, staged images and map, descriptor tables, pinned mirrors and bounce
// buffers.  Streams, events, the communicator and the tuning stay; the next call grows what it needs again.  The context must be the caller's
// (not in use by another thread); everything it has queued is waited for first.
int context_trim(rmgr_ssim_hip_Context* c)
{
    DeviceGuard device_guard_(c->device);
    if (device_guard_.rc) return device_guard_.rc;
    int rc = 0;
    if (!c->collectives.empty()) rc = comm_bounded_sync(c);
    else { const hipError_t e = hipStreamSynchronize(c->stream); if (e != hipSuccess) { (void)hipGetLastError(); rc = map_hip_error(e); } }
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->out_stream) (void)hipStreamSynchronize(c->out_stream);
    (void)hipGetLastError();
    struct Drop {
        static void dev(void* p) { if (p) (void)hipFree(p); }
        static void pin(void* p) { if (p) (void)hipHostFree(p); }
    };
    Drop::dev(c->partials); c->partials = NULL; c->partials_cap = 0;
    for (int i = 0; i < rmgr_ssim_hip_Context_::kDescSlots; ++i) {
        rmgr_ssim_hip_Context_::DescSlot& s = c->desc_slots[i];
        Drop::dev(s.dev); s.dev = NULL; s.dev_cap = 0;
        Drop::pin(s.host); s.host = NULL; s.host_cap = 0;
        s.live = 0; s.in_flight = false;
    }
    Drop::dev(c->stage_a); c->stage_a = NULL; c->stage_a_cap = 0;
    Drop::dev(c->stage_b); c->stage_b = NULL; c->stage_b_cap = 0;
    Drop::dev(c->stage_map); c->stage_map = NULL; c->stage_map_cap = 0;
    Drop::pin(c->h_sums); c->h_sums = NULL; c->h_sums_cap = 0;
    Drop::pin(c->h_stage); c->h_stage = NULL; c->h_stage_cap = 0;
    for (int i = 0; i < 2; ++i) {
        Drop::dev(c->slot_dev[i]); c->slot_dev[i] = NULL; c->slot_dev_cap[i] = 0;
        Drop::pin(c->slot_pin[i]); c->slot_pin[i] = NULL; c->slot_pin_cap[i] = 0;
        Drop::pin(c->h_map[i]); c->h_map[i] = NULL; c->h_map_cap[i] = 0;
    }
    Drop::dev(c->batch_sums); c->batch_sums = NULL; c->batch_sums_cap = 0;
    (void)hipGetLastError();
    return rc;
}

void pool_note_held_locked(rmgr_ssim_hip_Context* c, uint64_t dev, uint64_t pin)
{
    for (size_t i = 0; i < g_pool.held.size(); ++i)
        if (g_pool.held[i].c == c) { g_pool.held[i].device_bytes = dev; g_pool.held[i].pinned_bytes = pin; return; }
    try { const DefaultPool::Held h = {c, dev, pin}; g_pool.held.push_back(h); } catch (...) {}
}

// The lease ends: staging above the pool's retain cap ($RMGR_SSIM_HIP_POOL_RETAIN_MB, per context; default 256) goes back to the system
// before the context becomes available again -- the reference keeps nothing past the call (src/ssim.cpp:1048-1088) -- and what the
// context still holds is noted for rmgr_ssim_hip_get_default_pool_memory.
void pool_release(rmgr_ssim_hip_Context* c)
{
    uint64_t cap;
    { std::lock_guard<std::mutex> lk(g_pool.m); cap = g_pool.retain_cap; }
    uint64_t dev = 0, pin = 0;
    context_held(c, dev, pin);
    if (dev + pin > cap) {
        (void)context_trim(c);
        context_held(c, dev, pin);
    }
    {
        std::lock_guard<std::mutex> lk(g_pool.m);
        pool_note_held_locked(c, dev, pin);
        g_pool.idle.push_back(c);
    }
    g_pool.freed.notify_one();
}

// The context a call runs on: the caller's, or a leased default one (returned when the call ends).
struct Lease {
    rmgr_ssim_hip_Context* c;
    bool pooled;
    Lease() : c(NULL), pooled(false) {}
    int take(rmgr_ssim_hip_Context* given)
    {
        if (given) { c = given; return 0; }
        const int rc = pool_acquire(&c);
        if (rc) { c = NULL; return rc; }
        if (!c) return ENODEV;
        pooled = true;
        return 0;
    }
    ~Lease() { if (pooled && c) pool_release(c); }
private:
    Lease(const Lease&);
    Lease& operator=(const Lease&);
};

} // namespace

extern "C" {

rmgr_int32_t rmgr_ssim_hip_get_device_count(rmgr_int32_t* count) RMGR_NOEXCEPT
{
    if (!count) return EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_create(rmgr_ssim_hip_Context** out, rmgr_int32_t device, void* stream) RMGR_NOEXCEPT
{
    if (!out) return EINVAL;
    *out = NULL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return ENODEV; }
    if (device < 0 || device >= n) return EINVAL;
    DeviceGuard device_guard_(device);
    if (device_guard_.rc) return device_guard_.rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    rmgr_ssim_hip_Context* c = new (std::nothrow) rmgr_ssim_hip_Context_();
    if (!c) return ENOMEM;
    c->device = device;
    c->cu_count = prop.multiProcessorCount;
    c->xcd_count = 8;
    {
        int xccs = 0;
        if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, device) == hipSuccess && xccs >= 1) c->xcd_count = xccs;
        else (void)hipGetLastError();
    }
    c->stream = static_cast<hipStream_t>(stream);
    c->owns_stream = false;
    c->mode = RMGR_SSIM_HIP_MODE_EXACT;
    c->strip_rows = 0;
    c->variant = 0;
    c->partials = NULL; c->partials_cap = 0;
    memset(c->desc_slots, 0, sizeof(c->desc_slots));
    c->desc_next = 0;
    c->stage_a = NULL; c->stage_a_cap = 0;
    c->stage_b = NULL; c->stage_b_cap = 0;
    c->stage_map = NULL; c->stage_map_cap = 0;
    c->h_sums = NULL; c->h_sums_cap = 0;
    c->h_stage = NULL; c->h_stage_cap = 0;
    for (int i = 0; i < 2; ++i) { c->slot_dev[i] = c->slot_pin[i] = NULL; c->slot_dev_cap[i] = c->slot_pin_cap[i] = 0; c->slot_copied[i] = c->slot_done[i] = NULL; }
    c->batch_sums = NULL; c->batch_sums_cap = 0;
    c->copy_stream = NULL;
    c->out_stream = NULL;
    for (int i = 0; i < rmgr_ssim_hip_Context_::kMaxBands; ++i) c->band_copied[i] = c->band_done[i] = NULL;
    c->h_map[0] = c->h_map[1] = NULL; c->h_map_cap[0] = c->h_map_cap[1] = 0;
    c->map_ev[0] = c->map_ev[1] = NULL;
    c->comm = NULL;
    c->comm_nonblocking = false;
    c->comm_ranks = 0;
    c->profiling = false;
    c->clock_dev = NULL;
    c->wall_clock_khz = 100000;
    {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) c->wall_clock_khz = khz;
        else (void)hipGetLastError();
    }
    c->prof_launches = 0;
    c->prof_ms = 0.0;
    if (!stream) {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return map_hip_error(e); }
        c->owns_stream = true;
    }
    snprintf(c->describe, sizeof(c->describe), "%s %s, %d CUs in %d XCDs, %.0f MHz, %.1f GiB; rmgr-ssim hip backend (code object gfx950)",
             prop.name, prop.gcnArchName, prop.multiProcessorCount, c->xcd_count, prop.clockRate / 1000.0, prop.totalGlobalMem / 1073741824.0);
    *out = c;
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_destroy(rmgr_ssim_hip_Context* c) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    DeviceGuard device_guard_(c->device);
    if (!c->collectives.empty()) (void)comm_bounded_sync(c);      // the same bounded wait as rmgr_ssim_hip_synchronize (aborts a stuck collective)
    else (void)hipStreamSynchronize(c->stream);
    (void)rmgr_ssim_hip_comm_destroy(c);
    comm_forget_collectives(c);
    for (size_t i = 0; i < c->spare_events.size(); ++i) (void)hipEventDestroy(c->spare_events[i]);
    for (size_t i = 0; i < c->pending.size(); ++i) { (void)hipEventDestroy(c->pending[i].first); (void)hipEventDestroy(c->pending[i].second); }
    for (size_t i = 0; i < c->free_events.size(); ++i) { (void)hipEventDestroy(c->free_events[i].first); (void)hipEventDestroy(c->free_events[i].second); }
    if (c->partials) (void)hipFree(c->partials);
    if (c->clock_dev) (void)hipFree(c->clock_dev);
    for (int i = 0; i < rmgr_ssim_hip_Context_::kDescSlots; ++i) {
        if (c->desc_slots[i].dev) (void)hipFree(c->desc_slots[i].dev);
        if (c->desc_slots[i].host) (void)hipHostFree(c->desc_slots[i].host);
        if (c->desc_slots[i].used) (void)hipEventDestroy(c->desc_slots[i].used);
    }
    if (c->stage_a) (void)hipFree(c->stage_a);
    if (c->stage_b) (void)hipFree(c->stage_b);
    if (c->stage_map) (void)hipFree(c->stage_map);
    if (c->h_sums) (void)hipHostFree(c->h_sums);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (int i = 0; i < 2; ++i) {
        if (c->slot_dev[i]) (void)hipFree(c->slot_dev[i]);
        if (c->slot_pin[i]) (void)hipHostFree(c->slot_pin[i]);
        if (c->slot_copied[i]) (void)hipEventDestroy(c->slot_copied[i]);
        if (c->slot_done[i]) (void)hipEventDestroy(c->slot_done[i]);
    }
    if (c->batch_sums) (void)hipFree(c->batch_sums);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->out_stream) (void)hipStreamDestroy(c->out_stream);
    for (int i = 0; i < rmgr_ssim_hip_Context_::kMaxBands; ++i) {
        if (c->band_copied[i]) (void)hipEventDestroy(c->band_copied[i]);
        if (c->band_done[i]) (void)hipEventDestroy(c->band_done[i]);
    }
    for (int i = 0; i < 2; ++i) { if (c->h_map[i]) (void)hipHostFree(c->h_map[i]); if (c->map_ev[i]) (void)hipEventDestroy(c->map_ev[i]); }
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

// ctx == NULL set_mode / get_mode touch an int of the pool: when a default context exists they need no lease (they used to wait behind
// $RMGR_SSIM_HIP_POOL calls in flight, or create another context, to learn whether a device exists).  Only an EMPTY pool leases -- i.e. creates
// its first context -- to keep the contract that the call fails with ENODEV on a machine without a device (select_impl() returns 0 there).
namespace {
int default_pool_has_device()
{
    {
        std::lock_guard<std::mutex> guard(g_pool.m);
        pool_configure_locked();
        for (size_t i = 0; i < g_pool.all.size(); ++i)
            if (g_pool.all[i] != NULL) return 0;
    }
    Lease probe;
    return probe.take(NULL);
}
} // namespace

rmgr_int32_t rmgr_ssim_hip_set_mode(rmgr_ssim_hip_Context* c, rmgr_int32_t mode) RMGR_NOEXCEPT
{
    if (mode < RMGR_SSIM_HIP_MODE_EXACT || mode > RMGR_SSIM_HIP_MODE_SEPARABLE) return EINVAL;
    if (!c) {
        // the process-wide default contexts of the drop-in entry points (what rmgr::ssim::select_impl switches): a property of the
        // pool, applied to a context when a call leases it (calls already in flight keep the mode they were leased with).
        const int rc = default_pool_has_device();
        if (rc) return rc;
        std::lock_guard<std::mutex> guard(g_pool.m);
#if defined(RMGR_SSIM_USE_DOUBLE) && RMGR_SSIM_USE_DOUBLE
        if (mode == RMGR_SSIM_HIP_MODE_EXACT || mode == RMGR_SSIM_HIP_MODE_UNFUSED) mode = RMGR_SSIM_HIP_MODE_DOUBLE;   // a double build stays double
#endif
        g_pool.mode = mode;
        return 0;
    }
    c->mode = mode;
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_get_mode(const rmgr_ssim_hip_Context* c, rmgr_int32_t* mode) RMGR_NOEXCEPT
{
    if (!mode) return EINVAL;
    if (!c) {                 // the process-wide default contexts, as for set_mode (ENODEV without a device)
        const int rc = default_pool_has_device();
        if (rc) return rc;
        std::lock_guard<std::mutex> guard(g_pool.m);
        *mode = g_pool.mode;
        return 0;
    }
    *mode = c->mode;
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_set_tuning(rmgr_ssim_hip_Context* c, rmgr_int32_t stripRows, rmgr_int32_t variant) RMGR_NOEXCEPT
{
    if (!c || stripRows < 0 || variant < 0) return EINVAL;
    c->strip_rows = stripRows;
    c->variant = variant;
    return 0;
}

const char* rmgr_ssim_hip_get_kernel_source_id(void) RMGR_NOEXCEPT
{
    return ssim_hip::kernels_source_id();
}

rmgr_int32_t rmgr_ssim_hip_get_default_pool(rmgr_int32_t* contexts, rmgr_int32_t* limit) RMGR_NOEXCEPT
{
    std::lock_guard<std::mutex> lk(g_pool.m);
    pool_configure_locked();
    if (contexts) {
        *contexts = 0;
        for (size_t i = 0; i < g_pool.all.size(); ++i) *contexts += g_pool.all[i] != NULL;
    }
    if (limit) *limit = g_pool.limit;
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_get_default_pool_memory(rmgr_uint64_t* deviceBytes, rmgr_uint64_t* pinnedBytes, rmgr_uint64_t* retainCapBytes) RMGR_NOEXCEPT
{
    std::lock_guard<std::mutex> lk(g_pool.m);
    pool_configure_locked();
    uint64_t dev = 0, pin = 0;
    for (size_t i = 0; i < g_pool.held.size(); ++i) { dev += g_pool.held[i].device_bytes; pin += g_pool.held[i].pinned_bytes; }
    if (deviceBytes) *deviceBytes = dev;
    if (pinnedBytes) *pinnedBytes = pin;
    if (retainCapBytes) *retainCapBytes = g_pool.retain_cap;
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_trim_default_pool(void) RMGR_NOEXCEPT
{
    // the idle contexts leave the pool while they are trimmed (no call can lease one half-freed); contexts in use are left alone
    std::vector<rmgr_ssim_hip_Context*> mine;
    {
        std::lock_guard<std::mutex> lk(g_pool.m);
        mine.swap(g_pool.idle);
    }
    int rc = 0;
    for (size_t i = 0; i < mine.size(); ++i) {
        const int r = context_trim(mine[i]);
        if (r && !rc) rc = r;
    }
    {
        std::lock_guard<std::mutex> lk(g_pool.m);
        for (size_t i = 0; i < mine.size(); ++i) {
            uint64_t dev = 0, pin = 0;
            context_held(mine[i], dev, pin);
            pool_note_held_locked(mine[i], dev, pin);
            g_pool.idle.push_back(mine[i]);
        }
    }
    g_pool.freed.notify_all();
    return rc;
}

rmgr_int32_t rmgr_ssim_hip_trim(rmgr_ssim_hip_Context* c) RMGR_NOEXCEPT
{
    if (!c) return rmgr_ssim_hip_trim_default_pool();
    return context_trim(c);
}

rmgr_int32_t rmgr_ssim_hip_get_memory_info(const rmgr_ssim_hip_Context* c, rmgr_uint64_t* freeBytes, rmgr_uint64_t* totalBytes) RMGR_NOEXCEPT
{
    int device;
    if (c) device = c->device;
    else { std::lock_guard<std::mutex> lk(g_pool.m); pool_configure_locked(); device = g_pool.device; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return ENODEV; }
    if (device < 0 || device >= n) return EINVAL;
    DeviceGuard device_guard_(device);
    if (device_guard_.rc) return device_guard_.rc;
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (freeBytes) *freeBytes = f;
    if (totalBytes) *totalBytes = t;
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_get_abi_version(void) RMGR_NOEXCEPT
{
    return RMGR_SSIM_HIP_ABI_VERSION;
}

rmgr_int32_t rmgr_ssim_hip_get_plan(const rmgr_ssim_hip_Context* c, rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint32_t count, rmgr_ssim_hip_Plan* plan) RMGR_NOEXCEPT
{
    if (!plan || plan->structSize < RMGR_SSIM_HIP_PLAN_MIN_SIZE) return EINVAL;
    const int mode = c ? c->mode : RMGR_SSIM_HIP_MODE_EXACT, cus = c ? c->cu_count : 256, xcds = c ? c->xcd_count : 8, rows = c ? c->strip_rows : 0;
    int variant = c ? c->variant : 0;
    if (variant == 0 && rows == 0) variant = ssim_hip::default_variant(width, height, count, mode, cus);    // as enqueue() does
    const ssim_hip::Geometry geo = ssim_hip::plan(width, height, count, mode, rows, variant, cus, xcds);
    rmgr_ssim_hip_Plan full;
    memset(&full, 0, sizeof(full));
    full.structSize = plan->structSize;
    full.stripWidth = geo.strip_w;
    full.stripRows = geo.strip_rows;
    full.stripsX = geo.strips_x;
    full.stripsY = geo.strips_y;
    full.wavefronts = geo.strips_x * geo.strips_y * count;
    full.waveSlots = geo.wave_slots;
    full.earlyRowSums = ssim_hip::uses_early_row_sums(geo, mode, variant) ? 1u : 0u;
    full.cellRows = geo.cell_rows;
    full.cellsX = geo.cells_x;
    full.cellsY = geo.cells_y;
    full.balancedChunks = geo.n_chunks;
    full.balancedChunkRows = geo.chunk_cells * geo.cell_rows;
    full.balancedInterleave = geo.n_chunks ? geo.bal_stride : 0u;
    memcpy(plan, &full, std::min<size_t>(plan->structSize, sizeof(full)));      // never beyond what the caller allocated
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_enqueue_batch(rmgr_ssim_hip_Context* c, rmgr_uint32_t count, const rmgr_ssim_Params* params, double* sumsDevice) RMGR_NOEXCEPT
{
    if (!c || (count && (!params || !sumsDevice))) return EINVAL;
    if (count == 0) return 0;
    bool any_map = false;
    for (uint32_t i = 0; i < count; ++i) {
        if (params[i].imgA.topLeft == NULL || params[i].imgB.topLeft == NULL) return EINVAL;
        if (params[i].width != params[0].width || params[i].height != params[0].height) return EINVAL;
        any_map = any_map || params[i].ssimMap != NULL;
    }
    for (uint32_t i = 0; i < count; ++i)
        if (any_map && params[i].ssimMap == NULL) return EINVAL;   // all or none
    USE_DEVICE(c);
    // One launch covers up to 65535 pairs (grid.z); larger batches go out in consecutive launches.
    const uint32_t kMaxPerLaunch = 65535;
    PairDesc* descs = new (std::nothrow) PairDesc[std::min(kMaxPerLaunch, count)];
    if (!descs) return ENOMEM;
    int rc = 0;
    for (uint32_t first = 0; first < count && rc == 0; first += kMaxPerLaunch) {
        const uint32_t n = std::min(kMaxPerLaunch, count - first);
        for (uint32_t i = 0; i < n; ++i) descs[i] = make_desc(params[first + i]);
        rc = enqueue(c, params[0].width, params[0].height, n, descs, any_map, sumsDevice + first);
    }
    delete[] descs;
    return rc;
}

rmgr_int32_t rmgr_ssim_hip_enqueue_rows(rmgr_ssim_hip_Context* c, const rmgr_ssim_Params* params, rmgr_uint32_t yBegin, rmgr_uint32_t yRows, double* cellsDevice) RMGR_NOEXCEPT
{
    if (!c || !params || !cellsDevice) return EINVAL;
    if (params->imgA.topLeft == NULL || params->imgB.topLeft == NULL) return EINVAL;
    const uint32_t W = params->width, H = params->height, cell = ssim_hip::cell_rows_for(H);
    if (yBegin > H) return EINVAL;
    if (yRows == 0 || yBegin == H) return 0;                                      // an empty band (split_rows gives ranks beyond the image's cell rows (H, H)): nothing to do, whatever its alignment
    if ((yBegin % cell) != 0) return EINVAL;                                      // bands start on reduction-cell boundaries ...
    const uint32_t yEnd = yRows >= H - yBegin ? H : yBegin + yRows;
    if (yEnd != H && (yEnd % cell) != 0) return EINVAL;                           // ... and end on one, or at the image's last row
    if (W == 0 || yEnd == yBegin) return 0;
    USE_DEVICE(c);
    const PairDesc d = make_desc(*params);
    return enqueue(c, W, H, 1, &d, d.map != NULL, NULL, yBegin, yEnd - yBegin, false, cellsDevice);
}

rmgr_int32_t rmgr_ssim_hip_reduce_cells(rmgr_ssim_hip_Context* c, rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint32_t count,
                                        const double* cellsDevice, double* sumsDevice) RMGR_NOEXCEPT
{
    if (!c || (count && (!cellsDevice || !sumsDevice))) return EINVAL;
    if (count == 0) return 0;
    USE_DEVICE(c);
    const ssim_hip::Geometry geo = ssim_hip::plan(width, height, count, c->mode, c->strip_rows, c->variant, c->cu_count, c->xcd_count);
    int rc = grow_device(c->partials, c->partials_cap, ssim_hip::reduce_scratch_size(geo) + 1);     // the chunk sums of very large images
    if (rc) return rc;
    HIP_TRY(ssim_hip::launch_reduce(geo, cellsDevice, c->partials, sumsDevice, c->stream));
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_compute_ssim_batch_host(rmgr_ssim_hip_Context* c, rmgr_uint32_t count, const rmgr_ssim_Params* params, float* ssim) RMGR_NOEXCEPT
{
    if (count && (!params || !ssim)) return EINVAL;
    if (count == 0) return 0;
    for (uint32_t i = 0; i < count; ++i) {
        if (params[i].imgA.topLeft == NULL || params[i].imgB.topLeft == NULL || params[i].ssimMap != NULL) return EINVAL;
        if (params[i].width != params[0].width || params[i].height != params[0].height) return EINVAL;
    }
    int rc = 0;
    Lease lease;
    if ((rc = lease.take(c))) return rc;
    c = lease.c;
    USE_DEVICE(c);
    const uint32_t W = params[0].width, H = params[0].height;
    if (W == 0 || H == 0) {                       // 0/0, like the single call
        for (uint32_t i = 0; i < count; ++i) ssim[i] = mean_of(0.0, W, H);
        return 0;
    }
    if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        if (!c->slot_copied[i]) HIP_TRY(hipEventCreateWithFlags(&c->slot_copied[i], hipEventDisableTiming));
        if (!c->slot_done[i]) HIP_TRY(hipEventCreateWithFlags(&c->slot_done[i], hipEventDisableTiming));
    }
    if ((rc = grow_device(c->batch_sums, c->batch_sums_cap, count))) return rc;
    if ((rc = grow_pinned(c->h_sums, c->h_sums_cap, count))) return rc;

    const size_t kChunkBytes = size_t(48) << 20, kAlign = 256;
    const uint32_t kMaxChunkPairs = 4096;
    PairDesc* descs = new (std::nothrow) PairDesc[std::min(count, kMaxChunkPairs)];
    if (!descs) return ENOMEM;
    struct Free { PairDesc* p; ~Free() { delete[] p; } } free_descs = {descs};
    (void)free_descs;

    bool slot_busy[2] = {false, false};
    uint32_t first = 0;
    for (int k = 0; first < count; ++k) {
        const int slot = k & 1;
        // this chunk: as many pairs as fit the byte budget (at least one)
        uint32_t n = 0;
        size_t bytes = 0;
        bool small = true;
        while (first + n < count && n < kMaxChunkPairs) {
            int64_t loA, hiA, loB, hiB;
            extent(params[first + n].imgA, W, H, loA, hiA);
            extent(params[first + n].imgB, W, H, loB, hiB);
            const size_t nA = (size_t)(hiA - loA + 1), nB = (size_t)(hiB - loB + 1);
            const size_t need = ((nA + kAlign - 1) & ~(kAlign - 1)) + ((nB + kAlign - 1) & ~(kAlign - 1));
            if (n > 0 && bytes + need > kChunkBytes) break;
            bytes += need;
            small = small && (nA + nB <= kSmallStageBytes);
            ++n;
        }
        // the slot's previous occupant (chunk k-2) must have been consumed
        if (slot_busy[slot]) HIP_TRY(hipEventSynchronize(c->slot_done[slot]));
        if ((rc = grow_device(c->slot_dev[slot], c->slot_dev_cap[slot], bytes))) return rc;
        if (small && (rc = grow_pinned(c->slot_pin[slot], c->slot_pin_cap[slot], bytes))) return rc;
        size_t off = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const rmgr_ssim_Params& p = params[first + i];
            rmgr_ssim_Params dev = p;
            const rmgr_ssim_ImgParams* src[2] = {&p.imgA, &p.imgB};
            rmgr_ssim_ImgParams* dst[2] = {&dev.imgA, &dev.imgB};
            for (int j = 0; j < 2; ++j) {
                int64_t lo, hi;
                extent(*src[j], W, H, lo, hi);
                const size_t nb = (size_t)(hi - lo + 1);
                if (small) memcpy(c->slot_pin[slot] + off, src[j]->topLeft + lo, nb);
                else HIP_TRY(hipMemcpyAsync(c->slot_dev[slot] + off, src[j]->topLeft + lo, nb, hipMemcpyHostToDevice, c->copy_stream));
                dst[j]->topLeft = c->slot_dev[slot] + off - lo;
                off += (nb + kAlign - 1) & ~(kAlign - 1);
            }
            descs[i] = make_desc(dev);
        }
        if (small) HIP_TRY(hipMemcpyAsync(c->slot_dev[slot], c->slot_pin[slot], bytes, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(hipEventRecord(c->slot_copied[slot], c->copy_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->slot_copied[slot], 0));
        if ((rc = enqueue(c, W, H, n, descs, false, c->batch_sums + first))) return rc;
        HIP_TRY(hipEventRecord(c->slot_done[slot], c->stream));
        slot_busy[slot] = true;
        first += n;
    }
    HIP_TRY(hipMemcpyAsync(c->h_sums, c->batch_sums, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < count; ++i) ssim[i] = mean_of(c->h_sums[i], W, H);
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_finalize(rmgr_uint32_t count, const double* sums, rmgr_uint32_t width, rmgr_uint32_t height, float* ssim) RMGR_NOEXCEPT
{
    if (count && (!sums || !ssim)) return EINVAL;
    for (uint32_t i = 0; i < count; ++i) ssim[i] = mean_of(sums[i], width, height);
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_synchronize(rmgr_ssim_hip_Context* c) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    USE_DEVICE(c);
    if (!c->collectives.empty()) return comm_bounded_sync(c);      // an all-reduce is queued: its peers might never arrive
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_compute_ssim_device(rmgr_ssim_hip_Context* c, float* ssim, const rmgr_ssim_Params* params) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    int rc = validate(ssim, params, NULL);
    if (rc) return rc;
    USE_DEVICE(c);
    // The reduction kernel stores the sum straight into pinned host memory (mapped into the device's address
    // space): no device-to-host copy call on the latency path, just the stream synchronisation.
    if ((rc = grow_pinned(c->h_sums, c->h_sums_cap, 1))) return rc;
    const PairDesc d = make_desc(*params);
    if ((rc = enqueue(c, params->width, params->height, 1, &d, d.map != NULL, c->h_sums))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (ssim)
        *ssim = mean_of(c->h_sums[0], params->width, params->height);
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_compute_ssim_host(rmgr_ssim_hip_Context* c, float* ssim, const rmgr_ssim_Params* params,
                                             const rmgr_ssim_ThreadPool* threadPool) RMGR_NOEXCEPT
{
    int rc = validate(ssim, params, threadPool);
    if (rc) return rc;
    Lease lease;                         // ctx == NULL: one of the default contexts, for this call only (concurrent callers overlap)
    if ((rc = lease.take(c))) return rc;
    c = lease.c;
    USE_DEVICE(c);
    const uint32_t W = params->width, H = params->height;

    // The reference allocates its tile scratch through params->alloc exactly once and fails with
    // ENOMEM when that returns NULL (src/ssim.cpp:1048-1052).  The GPU path has no use for host
    // scratch, but the contract is kept observable: one 64-byte allocation, released before returning.
    void* user_mem = NULL;
    if (params->alloc != NULL) {
        user_mem = params->alloc(64, 64);        // nothing else needs host scratch: staging is pinned memory owned by the context
        if (user_mem == NULL) return ENOMEM;
    }
    struct Release {
        const rmgr_ssim_Params* p; void* m;
        ~Release() { if (m && p->dealloc) p->dealloc(m); }
    } rel = {params, user_mem};
    (void)rel;

    rmgr_ssim_Params dev = *params;
    const bool pooled = threadPool != NULL && threadPool->dispatch != NULL;
    bool staged = true;                  // false: the large-image copies are still to be issued
    int64_t loA = 0, hiA = 0, loB = 0, hiB = 0;
    if (W && H) {
        // Stage the byte range each image occupies; step/stride semantics carry over unchanged.
        extent(params->imgA, W, H, loA, hiA);
        extent(params->imgB, W, H, loB, hiB);
        const size_t nA = (size_t)(hiA - loA + 1), nB = (size_t)(hiB - loB + 1);
        if (nA + nB <= kSmallStageBytes) {
            // Small images: two pageable copies cost ~12 us each in driver overhead.  Gather both byte ranges in
            // one pinned buffer (a ~4 us memcpy at this size) and send them with a single DMA.
            const size_t offB = (nA + 63) & ~(size_t)63;
            if ((rc = grow_pinned(c->h_stage, c->h_stage_cap, offB + nB))) return rc;
            if ((rc = grow_device(c->stage_a, c->stage_a_cap, offB + nB))) return rc;
            if (!pooled) {                    // with the caller's thread pool the images are sent by its jobs (compute_via_pool())
                memcpy(c->h_stage, params->imgA.topLeft + loA, nA);
                memcpy(c->h_stage + offB, params->imgB.topLeft + loB, nB);
                HIP_TRY(hipMemcpyAsync(c->stage_a, c->h_stage, offB + nB, hipMemcpyHostToDevice, c->stream));
            }
            dev.imgA.topLeft = c->stage_a - loA;
            dev.imgB.topLeft = c->stage_a + offB - loB;
        } else {
            if ((rc = grow_device(c->stage_a, c->stage_a_cap, nA))) return rc;
            if ((rc = grow_device(c->stage_b, c->stage_b_cap, nB))) return rc;
            dev.imgA.topLeft = c->stage_a - loA;
            dev.imgB.topLeft = c->stage_b - loB;
            staged = false;                   // copied below: in one piece, or band by band
        }
        if (params->ssimMap) {
            if ((rc = grow_device(c->stage_map, c->stage_map_cap, (size_t)W * H))) return rc;
            dev.ssimMap = c->stage_map;      // dense W x H on the device
            dev.ssimStep = 1;
            dev.ssimStride = W;
        }
    } else {
        dev.ssimMap = NULL;
    }

    if ((rc = grow_pinned(c->h_sums, c->h_sums_cap, 1))) return rc;
    const PairDesc d = make_desc(dev);

    // A large pair whose map is wanted moves 2 B/px in and 4 B/px out over PCIe; the link is full duplex and the
    // 0.1 ms kernel is nothing next to either.  The image is therefore cut into row bands: band k+1 is copied in
    // while band k is computed (a row window of the same launch geometry; the cell-based reduction makes the sum
    // bit-identical to the one-launch result) and band k-1's map rows travel back, written by a helper thread
    // straight into the caller's buffer.
    // Without a map a few bands hide the kernel behind the copy, which pays only for very large images (compute_banded()).
    // An explicit $RMGR_SSIM_HIP_BANDS always takes the banded path (so that a band sweep measures what it says it does).
    const bool bands_forced = getenv("RMGR_SSIM_HIP_BANDS") != NULL;
    if (pooled)           // the caller brought a thread pool: its dispatch function runs the row-band jobs
        return compute_via_pool(c, ssim, *params, d, loA, loB, *threadPool);
    if (!staged && bandable(*params) && (params->ssimMap || bands_forced || (uint64_t)W * H >= (uint64_t(1) << 26))) {
        if ((rc = compute_banded(c, *params, dev, d, loA, loB))) return rc;
    } else {
        if (!staged) {
            HIP_TRY(hipMemcpyAsync(c->stage_a, params->imgA.topLeft + loA, (size_t)(hiA - loA + 1), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->stage_b, params->imgB.topLeft + loB, (size_t)(hiB - loB + 1), hipMemcpyHostToDevice, c->stream));
        }
        if ((rc = enqueue(c, W, H, 1, &d, d.map != NULL, c->h_sums))) return rc;      // sum lands in pinned host memory
        if (params->ssimMap && W && H) {
            if ((rc = map_rows_to_host(c, *params, 0, H, c->stream))) return rc;
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (ssim)
        *ssim = mean_of(c->h_sums[0], W, H);
    return 0;
}

// ---- one process, several devices ------------------------------------------------------------------------------------
// The reference parallelises one call over a caller-supplied thread pool (tile jobs, src/ssim.cpp:1048-1088; the OpenMP
// adapter src/ssim-openmp.c:26-47).  The batch-level counterpart here: a contiguous block of the pairs per device, one
// worker thread and one cached engine context per device slot, every block through the pipelined host-batch path above.
// Pairs are independent and every ssim[i] is written by exactly one worker, so there is no exchange step at all in one
// address space (the RCCL all-reduce exists for the one-process-per-GPU form, rmgr_ssim_hip_comm_*), and ssim[i] is
// bit-identical to the single-device call for any device list.
namespace {
std::mutex g_slots_lock;
std::vector<std::pair<int, rmgr_ssim_hip_Context*> > g_slots;     // (device, context) per slot of the last device lists; never freed
std::mutex g_multi_lock;                                          // one multi-device call at a time (the slots are shared)

rmgr_ssim_hip_Context* slot_context(size_t slot, int device, int* err)
{
    std::lock_guard<std::mutex> guard(g_slots_lock);
    try {
        if (g_slots.size() <= slot) g_slots.resize(slot + 1, std::make_pair(-1, (rmgr_ssim_hip_Context*)NULL));
    } catch (...) { *err = ENOMEM; return NULL; }
    if (g_slots[slot].second && g_slots[slot].first != device) {     // the slot served another device last time
        rmgr_ssim_hip_destroy(g_slots[slot].second);
        g_slots[slot].second = NULL;
    }
    if (!g_slots[slot].second) {
        *err = rmgr_ssim_hip_create(&g_slots[slot].second, device, NULL);
        if (*err) { g_slots[slot].second = NULL; return NULL; }
        g_slots[slot].first = device;
    }
    return g_slots[slot].second;
}
} // namespace

extern "C" rmgr_int32_t rmgr_ssim_hip_compute_ssim_batch_host_devices(const rmgr_int32_t* devices, rmgr_uint32_t deviceCount, rmgr_int32_t mode,
                                                                      rmgr_uint32_t count, const rmgr_ssim_Params* params, float* ssim) RMGR_NOEXCEPT
{
    if (count && (!params || !ssim)) return EINVAL;
    if (mode < RMGR_SSIM_HIP_MODE_EXACT || mode > RMGR_SSIM_HIP_MODE_SEPARABLE) return EINVAL;
    if (devices && deviceCount == 0) return EINVAL;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess) { (void)hipGetLastError(); visible = 0; }
    if (visible <= 0) return ENODEV;
    std::vector<int> devs;
    try {
        if (devices) devs.assign(devices, devices + deviceCount);
        else for (int d = 0; d < visible; ++d) devs.push_back(d);
    } catch (...) { return ENOMEM; }
    for (size_t k = 0; k < devs.size(); ++k)
        if (devs[k] < 0 || devs[k] >= visible) return EINVAL;
    if (count == 0) return 0;
    std::lock_guard<std::mutex> one_call(g_multi_lock);
    const size_t n = std::min<size_t>(devs.size(), count);          // no more workers than pairs
    // contiguous blocks, the first count % n slots one pair longer (ssim_amd/sharding.py split_batch)
    std::vector<int> rcs;
    std::vector<std::thread> workers;
    try { rcs.assign(n, 0); workers.reserve(n); } catch (...) { return ENOMEM; }
    struct Job {
        static void run(size_t slot, int device, int mode, uint32_t first, uint32_t cnt, const rmgr_ssim_Params* params, float* ssim, int* rc)
        {
            rmgr_ssim_hip_Context* c = slot_context(slot, device, rc);
            if (!c) return;
            if ((*rc = rmgr_ssim_hip_set_mode(c, mode))) return;
            *rc = rmgr_ssim_hip_compute_ssim_batch_host(c, cnt, params + first, ssim + first);
        }
    };
    const uint32_t base = (uint32_t)(count / n), extra = (uint32_t)(count % n);
    uint32_t first = 0;
    int launch_rc = 0;
    for (size_t k = 0; k < n; ++k) {
        const uint32_t cnt = base + (k < extra ? 1u : 0u);
        if (k + 1 == n) {
            Job::run(k, devs[k], mode, first, cnt, params, ssim, &rcs[k]);          // the calling thread takes the last block
        } else {
            try { workers.push_back(std::thread(Job::run, k, devs[k], (int)mode, first, cnt, params, ssim, &rcs[k])); }
            catch (...) { launch_rc = EAGAIN; break; }
        }
        first += cnt;
    }
    for (size_t k = 0; k < workers.size(); ++k) workers[k].join();
    if (launch_rc) return launch_rc;
    for (size_t k = 0; k < n; ++k)
        if (rcs[k]) return rcs[k];
    return 0;
}

} // extern "C"

namespace {

struct StagedPair {
    rmgr_ssim_hip_Context* c;
    Lease lease;
    DeviceGuard device;           // the context's device stays current for as long as the staged pair lives, i.e. for the WHOLE
                                  // entry point: its later allocations, events and launches must not land on the caller's device
    uint8_t* a; uint8_t* b;       // device copies of the two interleaved images, rows `pitch` bytes apart
    size_t pitch;
};

// Shared front half of the multi-channel entry points: context, validation, one H2D copy per image.
int stage_interleaved(rmgr_ssim_hip_Context* c, StagedPair& sp, const void* out1, const void* out2,
                      const uint8_t* imgA, ptrdiff_t strideA, const uint8_t* imgB, ptrdiff_t strideB,
                      uint32_t width, uint32_t height, uint32_t channels)
{
    if ((out1 == NULL && out2 == NULL) || imgA == NULL || imgB == NULL || channels == 0) return EINVAL;
    int rc = 0;
    if ((rc = sp.lease.take(c))) return rc;
    c = sp.lease.c;
    sp.c = c;
    if ((rc = sp.device.enter(c->device))) return rc;
    sp.pitch = ((size_t)width * channels + 3) & ~(size_t)3;      // dword-aligned rows for the packed luminance path
    const size_t bytes = sp.pitch * height + 4;
    if ((rc = grow_device(c->stage_a, c->stage_a_cap, bytes))) return rc;
    if ((rc = grow_device(c->stage_b, c->stage_b_cap, bytes))) return rc;
    sp.a = c->stage_a; sp.b = c->stage_b;
    if (width && height) {
        // rows may be stored bottom-up (negative stride): copy row 0 first either way
        HIP_TRY(hipMemcpy2DAsync(sp.a, sp.pitch, strideA >= 0 ? imgA : imgA + (ptrdiff_t)(height - 1) * strideA, (size_t)(strideA >= 0 ? strideA : -strideA),
                                 (size_t)width * channels, height, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpy2DAsync(sp.b, sp.pitch, strideB >= 0 ? imgB : imgB + (ptrdiff_t)(height - 1) * strideB, (size_t)(strideB >= 0 ? strideB : -strideB),
                                 (size_t)width * channels, height, hipMemcpyHostToDevice, c->stream));
    }
    return 0;
}

} // namespace

extern "C" rmgr_int32_t rmgr_ssim_hip_luminance_device(rmgr_ssim_hip_Context* c, rmgr_uint8_t* dstY, ptrdiff_t dstStride,
                                                       const rmgr_uint8_t* src, ptrdiff_t srcStep, ptrdiff_t srcStride,
                                                       rmgr_uint32_t width, rmgr_uint32_t height) RMGR_NOEXCEPT
{
    if (!c || !dstY || !src || srcStep < 3) return EINVAL;
    USE_DEVICE(c);
    HIP_TRY(ssim_hip::launch_luminance(dstY, dstStride, src, srcStep, srcStride, width, height, c->stream));
    return 0;
}

extern "C" rmgr_int32_t rmgr_ssim_hip_compute_ssim_channels_host(rmgr_ssim_hip_Context* ctx, float* ssim,
                                                                 const rmgr_uint8_t* imgA, ptrdiff_t strideA, const rmgr_uint8_t* imgB, ptrdiff_t strideB,
                                                                 rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint32_t channels, float* ssimMap) RMGR_NOEXCEPT
{
    StagedPair sp;
    int rc = stage_interleaved(ctx, sp, ssim, ssimMap, imgA, strideA, imgB, strideB, width, height, channels);
    if (rc) return rc;
    rmgr_ssim_hip_Context* c = sp.c;
    const bool bottomA = strideA < 0, bottomB = strideB < 0;
    const size_t mapFloats = (size_t)width * height * channels;
    if (ssimMap && mapFloats && (rc = grow_device(c->stage_map, c->stage_map_cap, mapFloats))) return rc;
    if ((rc = grow_pinned(c->h_sums, c->h_sums_cap, channels))) return rc;
    struct Descs {               // no exceptions in here: a failed allocation is ENOMEM, like everywhere else
        PairDesc* p;
        explicit Descs(size_t n) : p(new (std::nothrow) PairDesc[n]) {}
        ~Descs() { delete[] p; }
    } descs_owner(channels);
    PairDesc* const descs = descs_owner.p;
    if (!descs) return ENOMEM;
    for (uint32_t ch = 0; ch < channels; ++ch) {
        PairDesc& d = descs[ch];
        // the staged copies hold the rows in the order they were copied (row 0 of a bottom-up image last)
        d.a = (bottomA ? sp.a + (size_t)(height ? height - 1 : 0) * sp.pitch : sp.a) + ch; d.a_step = channels; d.a_stride = bottomA ? -(int64_t)sp.pitch : (int64_t)sp.pitch;
        d.b = (bottomB ? sp.b + (size_t)(height ? height - 1 : 0) * sp.pitch : sp.b) + ch; d.b_step = channels; d.b_stride = bottomB ? -(int64_t)sp.pitch : (int64_t)sp.pitch;
        d.map = (ssimMap && mapFloats) ? c->stage_map + ch : NULL;
        d.map_step = d.map ? channels : 0;
        d.map_stride = d.map ? (int64_t)width * channels : 0;
    }
    if ((rc = enqueue(c, width, height, channels, descs, ssimMap != NULL && mapFloats, c->h_sums))) return rc;
    if (ssimMap && mapFloats)
        HIP_TRY(hipMemcpyAsync(ssimMap, c->stage_map, sizeof(float) * mapFloats, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (ssim)
        for (uint32_t ch = 0; ch < channels; ++ch) ssim[ch] = mean_of(c->h_sums[ch], width, height);
    return 0;
}

extern "C" rmgr_int32_t rmgr_ssim_hip_compute_ssim_luminance_host(rmgr_ssim_hip_Context* ctx, float* ssim,
                                                                  const rmgr_uint8_t* imgA, ptrdiff_t strideA, const rmgr_uint8_t* imgB, ptrdiff_t strideB,
                                                                  rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint32_t channels, float* ssimMap) RMGR_NOEXCEPT
{
    if (channels < 3) return EINVAL;
    StagedPair sp;
    int rc = stage_interleaved(ctx, sp, ssim, ssimMap, imgA, strideA, imgB, strideB, width, height, channels);
    if (rc) return rc;
    rmgr_ssim_hip_Context* c = sp.c;
    // the Y planes live behind the staged RGB data of each image (rows in copy order: flip if bottom-up)
    const size_t ypitch = ((size_t)width + 3) & ~(size_t)3;
    const size_t rgbBytes = sp.pitch * height + 4, yBytes = ypitch * height + 4;
    // grow WITHOUT losing the staged pixels: allocate the Y planes separately in the map staging area
    const size_t mapFloats = (size_t)width * height;
    const size_t needFloats = (ssimMap ? mapFloats : 0) + (2 * yBytes + 3) / 4 + 4;
    if ((rc = grow_device(c->stage_map, c->stage_map_cap, needFloats))) return rc;
    (void)rgbBytes;
    uint8_t* ya = reinterpret_cast<uint8_t*>(c->stage_map + (ssimMap ? mapFloats : 0));
    ya += (4 - (reinterpret_cast<uintptr_t>(ya) & 3u)) & 3u;
    uint8_t* yb = ya + yBytes;
    yb += (4 - (reinterpret_cast<uintptr_t>(yb) & 3u)) & 3u;
    HIP_TRY(ssim_hip::launch_luminance(ya, (int64_t)ypitch, sp.a, channels, (int64_t)sp.pitch, width, height, c->stream));
    HIP_TRY(ssim_hip::launch_luminance(yb, (int64_t)ypitch, sp.b, channels, (int64_t)sp.pitch, width, height, c->stream));
    if ((rc = grow_pinned(c->h_sums, c->h_sums_cap, 1))) return rc;
    PairDesc d;
    const bool bottomA = strideA < 0, bottomB = strideB < 0;
    d.a = bottomA ? ya + (size_t)(height ? height - 1 : 0) * ypitch : ya; d.a_step = 1; d.a_stride = bottomA ? -(int64_t)ypitch : (int64_t)ypitch;
    d.b = bottomB ? yb + (size_t)(height ? height - 1 : 0) * ypitch : yb; d.b_step = 1; d.b_stride = bottomB ? -(int64_t)ypitch : (int64_t)ypitch;
    d.map = (ssimMap && mapFloats) ? c->stage_map : NULL;
    d.map_step = d.map ? 1 : 0;
    d.map_stride = d.map ? width : 0;
    if ((rc = enqueue(c, width, height, 1, &d, d.map != NULL, c->h_sums))) return rc;
    if (d.map)
        HIP_TRY(hipMemcpyAsync(ssimMap, c->stage_map, sizeof(float) * mapFloats, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (ssim) *ssim = mean_of(c->h_sums[0], width, height);
    return 0;
}

// ---- RCCL, loaded lazily so that single-GPU users carry no dependency on it ----
//
// Every step that can wait for something outside this process has a DEADLINE ($RMGR_SSIM_HIP_COMM_TIMEOUT_S, default
// 30 s): loading the library, the bootstrap behind ncclGetUniqueId, the rendezvous of ncclCommInitRank, the enqueue of the
// all-reduce, the teardown.  The reference turns a failed worker into a bounded ECHILD return (src/ssim.cpp:1094-1097);
// the GPU-era counterpart of "a worker failed" is "a rank never arrived", and that must come back as an errno
// (ETIMEDOUT) too, not as a hang.  Mechanics: library load, ncclGetUniqueId and the communicator's creation run on a helper
// thread the caller waits for with a timeout; the communicator is requested NON-BLOCKING (ncclCommInitRankConfig, blocking
// = 0), its state polled with ncclCommGetAsyncError and, past the deadline, torn down with ncclCommAbort by the helper.  The
// RCCL builds of this image (2.26.6, 2.27.7) run the rendezvous inside ncclCommInitRankConfig all the same (the helper's log
// shows the call returning only when the communicator is ready), so there the helper's timeout is the bound that acts: a
// helper that never returns is abandoned -- the process keeps one parked thread; nothing it owns lives on the caller's stack.
namespace {

struct Rccl {
    void* handle;
    char  path[256];          // what dlopen() resolved (diagnostics: rmgr_ssim_hip_comm_describe)
    int   version;
    ncclResult_t (*GetVersion)(int*);
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommInitRankConfig)(ncclComm_t*, int, ncclUniqueId, int, ncclConfig_t*);     // optional
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*);                               // optional
    ncclResult_t (*CommAbort)(ncclComm_t);                                                      // optional
    ncclResult_t (*CommFinalize)(ncclComm_t);                                                   // optional
    ncclResult_t (*CommCount)(const ncclComm_t, int*);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*CommDestroy)(ncclComm_t);
};

template <typename F> void sym(void* h, const char* name, F& f) { f = reinterpret_cast<F>(dlsym(h, name)); }

// Which librccl: $RMGR_SSIM_HIP_RCCL_LIB if set; else the SONAME first -- a process that already carries an RCCL (a host
// framework that bundles its own copy next to its own HIP runtime, e.g. a PyTorch wheel) must get THAT copy back, not a
// second RCCL bound to a second HIP runtime -- then the development name and ROCm's default location.
// NEVER waits: the first caller loads (dlopen + dlsym, on whichever thread that is -- normally a communicator helper); while that
// is in progress every other caller gets NULL (-> ENOSYS / "not loadable yet") instead of queueing behind a load that an
// abandoned helper might be stuck in (std::call_once made every later caller wait without a bound).
// path_override: $RMGR_SSIM_HIP_RCCL_LIB as the CALLER's thread read it (helpers do not call getenv: the host process -- Python --
// may be changing its environment concurrently).
enum { RCCL_UNLOADED = 0, RCCL_LOADING = 1, RCCL_READY = 2, RCCL_ABSENT = 3 };
Rccl             g_rccl_api;
std::atomic<int> g_rccl_state(RCCL_UNLOADED);

// wait_for_load (communicator helpers only -- their caller bounds THEM): sleep through another thread's load instead of reporting NULL.
Rccl* rccl(const char* path_override = NULL, bool use_env = true, bool wait_for_load = false)
{
    for (;;) {
        const int st = g_rccl_state.load(std::memory_order_acquire);
        if (st == RCCL_READY) return &g_rccl_api;
        if (st == RCCL_ABSENT) return NULL;
        if (st == RCCL_LOADING) {
            if (!wait_for_load) return NULL;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
            continue;
        }
        int expected = RCCL_UNLOADED;
        if (g_rccl_state.compare_exchange_strong(expected, RCCL_LOADING)) break;
    }
    Rccl& api = g_rccl_api;
    memset(&api, 0, sizeof(api));
    const char* names[] = {path_override ? path_override : (use_env ? getenv("RMGR_SSIM_HIP_RCCL_LIB") : NULL), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (size_t i = 0; i < sizeof(names) / sizeof(names[0]) && !api.handle; ++i)
        if (names[i] && names[i][0]) api.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (api.handle) {
        sym(api.handle, "ncclGetVersion", api.GetVersion);
        sym(api.handle, "ncclGetUniqueId", api.GetUniqueId);
        sym(api.handle, "ncclCommInitRank", api.CommInitRank);
        sym(api.handle, "ncclCommInitRankConfig", api.CommInitRankConfig);
        sym(api.handle, "ncclCommGetAsyncError", api.CommGetAsyncError);
        sym(api.handle, "ncclCommAbort", api.CommAbort);
        sym(api.handle, "ncclCommFinalize", api.CommFinalize);
        sym(api.handle, "ncclCommCount", api.CommCount);
        sym(api.handle, "ncclAllReduce", api.AllReduce);
        sym(api.handle, "ncclCommDestroy", api.CommDestroy);
        if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy || !api.CommCount) { dlclose(api.handle); api.handle = NULL; }
    }
    if (api.handle) {
        if (api.GetVersion) (void)api.GetVersion(&api.version);
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(api.AllReduce), &info) && info.dli_fname) snprintf(api.path, sizeof(api.path), "%s", info.dli_fname);
    }
    g_rccl_state.store(api.handle ? RCCL_READY : RCCL_ABSENT, std::memory_order_release);
    return api.handle ? &api : NULL;
}

int map_nccl(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess:          return 0;
    case ncclInvalidArgument:
    case ncclInvalidUsage:     return EINVAL;
    case ncclInProgress:       return ETIMEDOUT;     // only ever surfaces once a deadline has passed
    case ncclSystemError:
    case ncclUnhandledCudaError:
    case ncclInternalError:
    default:                   return ECHILD;
    }
}

typedef std::chrono::steady_clock Clock;

double comm_timeout_s()
{
    if (const char* e = getenv("RMGR_SSIM_HIP_COMM_TIMEOUT_S")) {
        const double v = atof(e);
        if (v > 0.0) return v;
    }
    return 30.0;
}

Clock::time_point deadline_from_now(double seconds)
{
    return Clock::now() + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(seconds));
}

// $RMGR_SSIM_HIP_COMM_DEBUG=1: the helper thread reports its steps on stderr (where a stalled bootstrap stalled).  The variable
// is read on the CALLER's thread (comm_debug_wanted) and travels in the job: helpers never call getenv.
bool comm_debug_wanted()
{
    const char* e = getenv("RMGR_SSIM_HIP_COMM_DEBUG");
    return e && atoi(e) != 0;
}

void comm_debug(bool on, const char* what, double seconds = -1.0)
{
    if (!on) return;
    if (seconds >= 0.0) fprintf(stderr, "[rmgr-ssim comm] %s (%.3f s)\n", what, seconds);
    else                fprintf(stderr, "[rmgr-ssim comm] %s\n", what);
    fflush(stderr);
}

double seconds_since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// A job for the helper thread.  Everything the helper touches lives in this heap block (shared with the caller through a
// shared_ptr), so a helper that outlives its caller's patience has nothing dangling to write to.
struct CommJob {
    std::mutex m;
    std::condition_variable cv;
    bool done;                     // the helper has finished (under m)
    bool abandoned;                // the caller has given up waiting (under m): whatever the helper still produces is the helper's to clean up
    std::atomic<bool> cancel;      // the polling loop's view of `abandoned`
    int rc;
    // inputs
    int device, rank_count, rank;
    bool want_id, nonblocking_ok;
    bool debug;                    // $RMGR_SSIM_HIP_COMM_DEBUG, read by the caller
    std::string rccl_path;         // $RMGR_SSIM_HIP_RCCL_LIB, read by the caller ("" = unset)
    ncclUniqueId id;
    // outputs
    ncclComm_t comm;
    bool nonblocking;
    CommJob() : done(false), abandoned(false), cancel(false), rc(0), device(0), rank_count(0), rank(0), want_id(false), nonblocking_ok(true), debug(false), comm(NULL), nonblocking(false) { memset(&id, 0, sizeof(id)); }
};

void comm_job_body(const std::shared_ptr<CommJob>& j)
{
    const Clock::time_point t0 = Clock::now();
    int rc = 0;
    comm_debug(j->debug, "helper: loading librccl");
    Rccl* r = rccl(j->rccl_path.empty() ? NULL : j->rccl_path.c_str(), false, true);      // may load (and page in) half a gigabyte of library; no getenv here
    comm_debug(j->debug, r ? r->path : "helper: no usable librccl", seconds_since(t0));
    if (!r) rc = ENOSYS;
    else if (j->want_id) { rc = map_nccl(r->GetUniqueId(&j->id)); comm_debug(j->debug, "helper: ncclGetUniqueId returned", seconds_since(t0)); }
    else if (hipSetDevice(j->device) != hipSuccess) { (void)hipGetLastError(); rc = ENODEV; }
    else {
        ncclResult_t res;
        if (j->nonblocking_ok && r->CommInitRankConfig && r->CommGetAsyncError && r->CommAbort) {
            ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
            cfg.blocking = 0;
            res = r->CommInitRankConfig(&j->comm, j->rank_count, j->id, j->rank, &cfg);
            char msg[96];
            snprintf(msg, sizeof(msg), "helper: ncclCommInitRankConfig(blocking = 0) returned %d", (int)res);
            comm_debug(j->debug, msg, seconds_since(t0));
            // An error here is final (no second attempt with the plain call: the id's rendezvous has been used).
            if (res == ncclSuccess || res == ncclInProgress) {
                j->nonblocking = true;
                for (;;) {                                   // the rendezvous proceeds on RCCL's own thread
                    ncclResult_t state = ncclSuccess;
                    const ncclResult_t g = r->CommGetAsyncError(j->comm, &state);
                    if (g != ncclSuccess) { res = g; break; }
                    if (state != ncclInProgress) { res = state; break; }
                    if (j->cancel.load()) { res = ncclInProgress; break; }      // the caller's deadline passed
                    std::this_thread::sleep_for(std::chrono::microseconds(200));
                }
                comm_debug(j->debug, res == ncclSuccess ? "helper: communicator ready" : res == ncclInProgress ? "helper: deadline passed, aborting the communicator" : "helper: init failed, aborting the communicator", seconds_since(t0));
                if (res != ncclSuccess && j->comm) { (void)r->CommAbort(j->comm); j->comm = NULL; comm_debug(j->debug, "helper: ncclCommAbort returned", seconds_since(t0)); }
            }
        } else {
            res = r->CommInitRank(&j->comm, j->rank_count, j->id, j->rank);
            comm_debug(j->debug, "helper: ncclCommInitRank returned", seconds_since(t0));
        }
        if (res != ncclSuccess) j->comm = NULL;
        rc = map_nccl(res);
    }
    std::unique_lock<std::mutex> lk(j->m);
    j->rc = rc;
    j->done = true;
    const bool orphan = j->abandoned && j->comm != NULL;     // finished after the caller left: nobody will ever own this communicator
    lk.unlock();
    j->cv.notify_all();
    if (orphan) {
        if (r && r->CommAbort) (void)r->CommAbort(j->comm);
        j->comm = NULL;
        comm_debug(j->debug, "helper: late communicator aborted", seconds_since(t0));
    }
}

// Runs the job on a helper thread and waits for it until the deadline.  Past it the caller returns ETIMEDOUT AT ONCE: the
// helper is told to give up -- its polling loop aborts the half-built communicator, which can itself take seconds while
// RCCL's bootstrap thread winds down -- and is left to finish that on its own (detached; it owns everything it touches).
int run_comm_job(const std::shared_ptr<CommJob>& j, double timeout_s)
{
    j->debug = comm_debug_wanted();                           // the environment is read HERE, on the caller's thread
    if (const char* e = getenv("RMGR_SSIM_HIP_RCCL_LIB")) { try { j->rccl_path = e; } catch (...) { return ENOMEM; } }
    std::thread t;
    try { t = std::thread(comm_job_body, j); } catch (...) { return EAGAIN; }
    std::unique_lock<std::mutex> lk(j->m);
    if (j->cv.wait_until(lk, deadline_from_now(timeout_s), [&] { return j->done; })) { lk.unlock(); t.join(); return j->rc; }
    j->abandoned = true;
    j->cancel.store(true);
    lk.unlock();
    t.detach();
    return ETIMEDOUT;
}

// Waits for a non-blocking communicator's last call to leave the "in progress" state.
int comm_wait_ready(Rccl* r, rmgr_ssim_hip_Context* c, ncclResult_t first)
{
    if (first != ncclInProgress) return map_nccl(first);
    if (!c->comm_nonblocking || !r->CommGetAsyncError) return map_nccl(first);
    const Clock::time_point deadline = deadline_from_now(comm_timeout_s());
    for (;;) {
        ncclResult_t state = ncclSuccess;
        const ncclResult_t g = r->CommGetAsyncError(c->comm, &state);
        if (g != ncclSuccess) return map_nccl(g);
        if (state != ncclInProgress) return map_nccl(state);
        if (Clock::now() > deadline) return ETIMEDOUT;
        std::this_thread::yield();
    }
}

void comm_abort(Rccl* r, rmgr_ssim_hip_Context* c);

// rmgr_ssim_hip_synchronize() / _destroy() of a context that owns a communicator: the stream may hold an all-reduce whose peers never
// launch theirs, and hipStreamSynchronize() would then wait forever.  So the wait POLLS, and it bounds the collectives only: each
// queued all-reduce sits between two events (comm_allreduce_sums); while the oldest outstanding one has not had its turn (its
// `begin` event is not complete) ordinary work is running and no clock ticks; from the moment `begin` completes that collective
// has $RMGR_SSIM_HIP_COMM_TIMEOUT_S to finish; past that the communicator is aborted (ncclCommAbort releases the kernel that spins
// on the missing peers), the wait for that kernel to leave is itself bounded by another such interval, and the caller gets
// ETIMEDOUT.  Without ncclCommAbort nothing can release the kernel: ETIMEDOUT is returned with the stream left as it is.  The
// poll sleeps (50 us, doubling up to 1 ms) instead of spinning.
hipEvent_t comm_take_event(rmgr_ssim_hip_Context* c)
{
    hipEvent_t e = NULL;
    if (!c->spare_events.empty()) { e = c->spare_events.back(); c->spare_events.pop_back(); return e; }
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return NULL; }
    return e;
}

void comm_forget_collectives(rmgr_ssim_hip_Context* c)
{
    while (!c->collectives.empty()) {
        c->spare_events.push_back(c->collectives.front().begin);
        c->spare_events.push_back(c->collectives.front().end);
        c->collectives.pop_front();
    }
}

int comm_bounded_sync(rmgr_ssim_hip_Context* c)
{
    const double limit = comm_timeout_s();
    std::chrono::microseconds nap(50);
    for (;;) {
        while (!c->collectives.empty()) {                        // retire what has completed
            const hipError_t e = hipEventQuery(c->collectives.front().end);
            if (e == hipErrorNotReady) break;
            if (e != hipSuccess) { (void)hipGetLastError(); return map_hip_error(e); }
            c->spare_events.push_back(c->collectives.front().begin);
            c->spare_events.push_back(c->collectives.front().end);
            c->collectives.pop_front();
        }
        if (c->collectives.empty()) {                            // nothing left on the stream can wait for a peer
            HIP_TRY(hipStreamSynchronize(c->stream));
            return 0;
        }
        rmgr_ssim_hip_Context_::Collective& k = c->collectives.front();
        if (!k.started) {
            const hipError_t e = hipEventQuery(k.begin);
            if (e == hipSuccess) { k.started = true; k.since = Clock::now(); }
            else if (e != hipErrorNotReady) { (void)hipGetLastError(); return map_hip_error(e); }
        } else if (std::chrono::duration<double>(Clock::now() - k.since).count() > limit) {
            Rccl* r = rccl();
            const bool can_abort = r && r->CommAbort;
            const hipEvent_t stuck = k.end;
            comm_abort(r, c);
            if (can_abort) {                                    // the aborted collective's kernel exits: wait for it, bounded
                const Clock::time_point until = deadline_from_now(limit);
                while (hipEventQuery(stuck) == hipErrorNotReady && Clock::now() < until) std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            (void)hipGetLastError();
            comm_forget_collectives(c);
            return ETIMEDOUT;
        }
        std::this_thread::sleep_for(nap);
        if (nap < std::chrono::microseconds(1000)) nap *= 2;
    }
}

void comm_abort(Rccl* r, rmgr_ssim_hip_Context* c)
{
    if (!c->comm) return;
    if (r && r->CommAbort) (void)r->CommAbort(c->comm);
    c->comm = NULL;               // without ncclCommAbort the handle is leaked rather than destroyed: ncclCommDestroy would wait for the peers
    c->comm_nonblocking = false;
    c->comm_ranks = 0;
}

} // namespace

extern "C" rmgr_int32_t rmgr_ssim_hip_comm_get_unique_id(unsigned char id[RMGR_SSIM_HIP_COMM_ID_BYTES]) RMGR_NOEXCEPT
{
    static_assert(sizeof(ncclUniqueId) == RMGR_SSIM_HIP_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id) return EINVAL;
    std::shared_ptr<CommJob> j;
    try { j = std::make_shared<CommJob>(); } catch (...) { return ENOMEM; }
    j->want_id = true;
    const int rc = run_comm_job(j, comm_timeout_s());
    if (rc == 0) memcpy(id, &j->id, sizeof(j->id));
    return rc;
}

extern "C" rmgr_int32_t rmgr_ssim_hip_comm_init(rmgr_ssim_hip_Context* c, const unsigned char id[RMGR_SSIM_HIP_COMM_ID_BYTES],
                                                rmgr_int32_t rankCount, rmgr_int32_t rank) RMGR_NOEXCEPT
{
    if (!c || !id || rankCount < 1 || rank < 0 || rank >= rankCount || c->comm) return EINVAL;
    std::shared_ptr<CommJob> j;
    try { j = std::make_shared<CommJob>(); } catch (...) { return ENOMEM; }
    j->device = c->device;
    j->rank_count = rankCount;
    j->rank = rank;
    if (const char* e = getenv("RMGR_SSIM_HIP_COMM_BLOCKING")) j->nonblocking_ok = atoi(e) == 0;
    memcpy(&j->id, id, sizeof(j->id));
    const int rc = run_comm_job(j, comm_timeout_s());
    if (rc) return rc;
    c->comm = j->comm;
    c->comm_nonblocking = j->nonblocking;
    c->comm_ranks = 0;
    Rccl* r = rccl();
    int n = 0;
    if (r && r->CommCount(c->comm, &n) == ncclSuccess) c->comm_ranks = n;
    return 0;
}

extern "C" rmgr_int32_t rmgr_ssim_hip_comm_rank_count(const rmgr_ssim_hip_Context* c, rmgr_int32_t* rankCount) RMGR_NOEXCEPT
{
    if (!c || !rankCount) return EINVAL;
    *rankCount = c->comm ? c->comm_ranks : 0;      // what RCCL itself reports for the communicator (ncclCommCount), 0 without one
    return 0;
}

extern "C" rmgr_int32_t rmgr_ssim_hip_comm_allreduce_sums(rmgr_ssim_hip_Context* c, double* sumsDevice, rmgr_uint32_t count) RMGR_NOEXCEPT
{
    if (!c || !c->comm || (count && !sumsDevice)) return EINVAL;
    if (count == 0) return 0;
    Rccl* r = rccl();
    if (!r) return ENOSYS;
    USE_DEVICE(c);
    rmgr_ssim_hip_Context_::Collective k;
    k.begin = comm_take_event(c); k.end = comm_take_event(c); k.started = false;
    if (!k.begin || !k.end) { if (k.begin) c->spare_events.push_back(k.begin); return ENOMEM; }
    if (hipEventRecord(k.begin, c->stream) != hipSuccess) {
        (void)hipGetLastError();
        c->spare_events.push_back(k.begin); c->spare_events.push_back(k.end);
        return ECHILD;
    }
    const int rc = comm_wait_ready(r, c, r->AllReduce(sumsDevice, sumsDevice, count, ncclFloat64, ncclSum, c->comm, c->stream));
    if (rc == ETIMEDOUT) comm_abort(r, c);         // the enqueue itself never completed: the communicator is gone
    if (rc == 0 && hipEventRecord(k.end, c->stream) == hipSuccess) {
        c->collectives.push_back(k);               // rmgr_ssim_hip_synchronize bounds its wait for THIS collective (comm_bounded_sync)
        while (c->collectives.size() > 64) {       // a caller that never synchronises: the oldest have long completed or will be caught by the next
            if (hipEventQuery(c->collectives.front().end) != hipSuccess) { (void)hipGetLastError(); break; }
            c->spare_events.push_back(c->collectives.front().begin); c->spare_events.push_back(c->collectives.front().end);
            c->collectives.pop_front();
        }
    } else {
        (void)hipGetLastError();
        c->spare_events.push_back(k.begin); c->spare_events.push_back(k.end);
    }
    return rc;
}

extern "C" rmgr_int32_t rmgr_ssim_hip_comm_destroy(rmgr_ssim_hip_Context* c) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    if (!c->comm) return 0;
    Rccl* r = rccl();
    if (!r) { c->comm = NULL; return ENOSYS; }
    USE_DEVICE(c);
    int rc = 0;
    if (!c->collectives.empty()) {                 // queued all-reduces first, with their deadline: a stuck one aborts the communicator
        rc = comm_bounded_sync(c);
        if (!c->comm) return rc;
    }
    if (c->comm_nonblocking && r->CommFinalize) {
        // ncclCommFinalize is the asynchronous half of the teardown; ncclCommDestroy then only frees
        rc = comm_wait_ready(r, c, r->CommFinalize(c->comm));
        if (rc == ETIMEDOUT) { comm_abort(r, c); return rc; }
    }
    const int rd = map_nccl(r->CommDestroy(c->comm));
    c->comm = NULL;
    c->comm_nonblocking = false;
    c->comm_ranks = 0;
    return rc ? rc : rd;
}

extern "C" const char* rmgr_ssim_hip_comm_describe(void) RMGR_NOEXCEPT
{
    static char text[384];
    static std::mutex m;
    std::lock_guard<std::mutex> lk(m);
    Rccl* r = rccl();
    if (!r) snprintf(text, sizeof(text), "rccl: not loadable (%s)", dlerror() ? "dlopen failed" : "no usable librccl");
    else snprintf(text, sizeof(text), "rccl %d.%d.%d from %s; non-blocking init %s; deadline %.1f s", r->version / 10000, (r->version / 100) % 100, r->version % 100,
                  r->path[0] ? r->path : "?", (r->CommInitRankConfig && r->CommGetAsyncError && r->CommAbort) ? "available" : "unavailable", comm_timeout_s());
    return text;
}

extern "C" {

rmgr_int32_t rmgr_ssim_hip_malloc(rmgr_ssim_hip_Context* c, void** p, size_t size) RMGR_NOEXCEPT
{
    if (!c || !p) return EINVAL;
    USE_DEVICE(c);
    HIP_TRY(hipMalloc(p, size ? size : 1));
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_free(rmgr_ssim_hip_Context* c, void* p) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    USE_DEVICE(c);
    HIP_TRY(hipFree(p));
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_memcpy_h2d(rmgr_ssim_hip_Context* c, void* dst, const void* src, size_t size) RMGR_NOEXCEPT
{
    if (!c || (size && (!dst || !src))) return EINVAL;
    USE_DEVICE(c);
    HIP_TRY(hipMemcpyAsync(dst, src, size, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_memcpy_d2h(rmgr_ssim_hip_Context* c, void* dst, const void* src, size_t size) RMGR_NOEXCEPT
{
    if (!c || (size && (!dst || !src))) return EINVAL;
    USE_DEVICE(c);
    HIP_TRY(hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

namespace {
// The per-XCD clock counters (ssim_kernels.hip clock_begin): mean and lowest shader clock over the XCDs that reported, and the launches counted (per XCD, the most any saw).
void clocks_from(const uint64_t* v, int xcds, int wall_clock_khz, double* mean_mhz, double* min_mhz, rmgr_uint64_t* launches)
{
    double sum = 0.0, lo = 0.0;
    int n = 0;
    uint64_t most = 0;
    for (int x = 0; x < xcds && x < (int)ssim_hip::kClockMaxXcds; ++x) {
        const uint64_t* c = v + ssim_hip::kClockStride * x;
        if (c[3] == 0) continue;
        const double mhz = (double)c[2] / (double)c[3] * (double)wall_clock_khz / 1000.0;
        sum += mhz;
        lo = n == 0 ? mhz : std::min(lo, mhz);
        most = std::max(most, c[4]);
        ++n;
    }
    if (mean_mhz) *mean_mhz = n ? sum / n : 0.0;
    if (min_mhz) *min_mhz = lo;
    if (launches) *launches = most;
}

// Reads and clears the clock counters the profiled launches added to (after the stream is idle).
int read_clock(rmgr_ssim_hip_Context* c, double* mhz, double* min_mhz, rmgr_uint64_t* launches)
{
    if (mhz) *mhz = 0.0;
    if (min_mhz) *min_mhz = 0.0;
    if (launches) *launches = 0;
    if (!c->clock_dev) return 0;
    uint64_t v[ssim_hip::kClockWords];
    HIP_TRY(hipMemcpyAsync(v, c->clock_dev, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemsetAsync(c->clock_dev, 0, sizeof(v), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    clocks_from(v, c->xcd_count, c->wall_clock_khz, mhz, min_mhz, launches);
    return 0;
}
} // namespace

rmgr_int32_t rmgr_ssim_hip_set_profiling(rmgr_ssim_hip_Context* c, rmgr_int32_t enabled) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    if (enabled && !c->clock_dev) {
        USE_DEVICE(c);
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->clock_dev), ssim_hip::kClockWords * sizeof(uint64_t)));
        HIP_TRY(hipMemsetAsync(c->clock_dev, 0, ssim_hip::kClockWords * sizeof(uint64_t), c->stream));
    }
    c->profiling = enabled != 0;
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_get_profile_clock(rmgr_ssim_hip_Context* c, double* shaderMHz, double* slowestXcdMHz, rmgr_uint64_t* launches) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    USE_DEVICE(c);
    return read_clock(c, shaderMHz, slowestXcdMHz, launches);
}

rmgr_int32_t rmgr_ssim_hip_get_profile(rmgr_ssim_hip_Context* c, rmgr_uint64_t* launches, double* kernelMs) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    USE_DEVICE(c);
    int rc = drain_profile(c);
    if (rc) return rc;
    if (launches) *launches = c->prof_launches;
    if (kernelMs) *kernelMs = c->prof_ms;
    c->prof_launches = 0;
    c->prof_ms = 0.0;
    return 0;
}

// ---- rmgr_ssim_hip_tune: the plan for one launch shape, MEASURED on the device the context runs on ---------------------------------------------
// plan()'s default is a model fitted on 256-CU boxes that differ by +-4 % (strip height by a packing model, EARLY by launch length, the balanced
// schedule by a priced rule); this times the handful of candidates that model chooses between -- on THIS device, at THIS clock -- and keeps the
// winner for later launches of the shape.  Candidates: the default; the two-column strips at the default height with the row sums in the blur
// phase / EARLY; the default kernel at half and at twice the strip height; the balanced schedule (no map; modes 0, 3, 1); the one-column kernel for
// small launches.  Every candidate gives the same bits (cells at absolute positions, fixed-order reduction): only time is at stake.
namespace {

struct TuneCandidate { int variant, rows; double ms; uint64_t key; };

uint64_t plan_key(const ssim_hip::Geometry& g, bool early, int one_column)
{
    uint64_t k = g.strip_w;
    k = k * 1000003u + g.strip_rows; k = k * 1000003u + g.n_chunks; k = k * 1000003u + g.chunk_cells; k = k * 1000003u + g.bal_stride;
    return (k * 4 + (early ? 1 : 0)) * 2 + (uint64_t)one_column;
}

} // namespace

rmgr_int32_t rmgr_ssim_hip_tune(rmgr_ssim_hip_Context* c, rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint32_t count, rmgr_int32_t withMap,
                                rmgr_ssim_hip_TuneResult* result) RMGR_NOEXCEPT
{
    if (!c || width == 0 || height == 0 || count == 0 || count > 65535u) return EINVAL;
    if (result && result->structSize < RMGR_SSIM_HIP_TUNE_RESULT_MIN_SIZE) return EINVAL;
    USE_DEVICE(c);
    const bool map = withMap != 0;
    const int mode = c->mode;
    // forget an earlier choice for this shape: the default candidate must run the default
    for (size_t i = 0; i < c->tuned.size(); ++i)
        if (c->tuned[i].width == width && c->tuned[i].height == height && c->tuned[i].count == count && c->tuned[i].mode == mode && c->tuned[i].map == map) { c->tuned.erase(c->tuned.begin() + i); break; }

    // candidates, de-duplicated by the launch they produce
    const int v0 = ssim_hip::default_variant(width, height, count, mode, c->cu_count);
    const ssim_hip::Geometry g0 = ssim_hip::plan(width, height, count, mode, 0, v0, c->cu_count, c->xcd_count);
    const uint32_t cell = g0.cell_rows, R = g0.strip_rows;
    std::vector<TuneCandidate> cand;
    try {
        struct Add {
            static void one(std::vector<TuneCandidate>& list, rmgr_ssim_hip_Context* c, uint32_t w, uint32_t h, uint32_t n, bool map, int variant, int rows)
            {
                int v = variant;
                if (v == 0 && rows == 0) v = ssim_hip::default_variant(w, h, n, c->mode, c->cu_count);
                ssim_hip::Geometry g = ssim_hip::plan(w, h, n, c->mode, rows, v, c->cu_count, c->xcd_count);
                if (map) { g.chunk_cells = 0; g.n_chunks = 0; g.bal_stride = 1; }                   // launches with a map run the strips
                if (ssim_hip::is_balanced_variant(variant) && g.n_chunks == 0) return;             // no balanced form for this launch
                const bool one = c->mode == RMGR_SSIM_HIP_MODE_DOUBLE || v == 1;
                const TuneCandidate t = {variant, rows, 0.0, plan_key(g, g.n_chunks ? true : ssim_hip::uses_early_row_sums(g, c->mode, v), one ? 1 : 0)};
                for (size_t i = 0; i < list.size(); ++i) if (list[i].key == t.key) return;
                list.push_back(t);
            }
        };
        Add::one(cand, c, width, height, count, map, 0, 0);                                        // the default, first
        if (mode != RMGR_SSIM_HIP_MODE_DOUBLE) {
            Add::one(cand, c, width, height, count, map, 2, (int)R);
            if (mode == RMGR_SSIM_HIP_MODE_EXACT || mode == RMGR_SSIM_HIP_MODE_UNFUSED) Add::one(cand, c, width, height, count, map, 3, (int)R);
            if (!map) Add::one(cand, c, width, height, count, map, 6, 0);
            if ((uint64_t)width * height * count <= (uint64_t(1) << 22)) Add::one(cand, c, width, height, count, map, 1, 0);
        }
        const uint32_t half = std::max(cell, ((R / 2) + cell - 1) & ~(cell - 1)), twice = std::min<uint32_t>(2 * R, (height + cell - 1) & ~(cell - 1));
        const int keep = (mode == RMGR_SSIM_HIP_MODE_DOUBLE) ? 0 : (g0.strip_w == 64 ? 1 : ssim_hip::uses_early_row_sums(g0, mode, v0) ? 3 : 2);
        if (!map && g0.n_chunks == 0) {            // where the chunks would divide the strip column evenly, the strips at the chunk height: the same partition without the segment loop
            const ssim_hip::Geometry gb = ssim_hip::plan(width, height, count, mode, 0, 6, c->cu_count, c->xcd_count);
            if (gb.n_chunks && gb.cells_y % gb.chunk_cells == 0) Add::one(cand, c, width, height, count, map, keep ? keep : 2, (int)(gb.chunk_cells * cell));
        } else if (!map && g0.cells_y % g0.chunk_cells == 0) {
            Add::one(cand, c, width, height, count, map, (mode == RMGR_SSIM_HIP_MODE_EXACT || mode == RMGR_SSIM_HIP_MODE_UNFUSED) ? 3 : 2, (int)(g0.chunk_cells * cell));
        }
        if (half != R) Add::one(cand, c, width, height, count, map, keep, (int)half);
        if (twice != R) Add::one(cand, c, width, height, count, map, keep, (int)twice);
    } catch (...) { return ENOMEM; }

    // synthetic pairs of the shape (SURVEY.md 8(d) pattern): distinct images up to ~1.5 GB, then the descriptors cycle through them
    const size_t plane = (size_t)width * height, per_pair = 2 * plane + (map ? 4 * plane : 0);
    const uint32_t distinct = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(count, (uint64_t(3) << 29) / per_pair));
    uint8_t* images = NULL;
    float* maps = NULL;
    double* sums = NULL;
    PairDesc* descs = new (std::nothrow) PairDesc[count];
    if (!descs) return ENOMEM;
    struct Cleanup {
        uint8_t*& images; float*& maps; double*& sums; PairDesc* descs;
        ~Cleanup() { if (images) (void)hipFree(images); if (maps) (void)hipFree(maps); if (sums) (void)hipFree(sums); delete[] descs; (void)hipGetLastError(); }
    } cleanup = {images, maps, sums, descs};
    (void)cleanup;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&images), 2 * plane * distinct));
    if (map) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&maps), sizeof(float) * plane * distinct));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sums), sizeof(double) * count));
    for (uint32_t i = 0; i < distinct; ++i)
        HIP_TRY(ssim_hip::launch_synth_pair(images + 2 * plane * i, width, images + 2 * plane * i + plane, width, width, height, 0x5EEDull + i, c->stream));
    for (uint32_t i = 0; i < count; ++i) {
        const uint32_t k = i % distinct;
        PairDesc& d = descs[i];
        d.a = images + 2 * plane * k; d.a_step = 1; d.a_stride = width;
        d.b = d.a + plane;            d.b_step = 1; d.b_stride = width;
        d.map = map ? maps + plane * k : NULL; d.map_step = map ? 1 : 0; d.map_stride = map ? (int64_t)width : 0;
    }

    // timing: rounds of (every candidate: one untimed + three timed launches), candidates interleaved so that clock drift hits all alike; the
    // figure of a candidate is the median of its per-round means
    const int saved_rows = c->strip_rows, saved_variant = c->variant;
    const bool saved_prof = c->profiling;
    int rc = drain_profile(c);
    const uint64_t saved_launches = c->prof_launches;
    const double saved_ms = c->prof_ms;
    const int rounds = 3, per_round = 3;
    std::vector<std::vector<double> > samples(cand.size());
    for (int r = 0; r < rounds && !rc; ++r) {
        for (size_t k = 0; k < cand.size() && !rc; ++k) {
            c->strip_rows = cand[k].rows; c->variant = cand[k].variant;
            c->profiling = false;
            rc = enqueue(c, width, height, count, descs, map, sums);
            c->profiling = true;
            c->prof_launches = 0; c->prof_ms = 0.0;
            for (int j = 0; j < per_round && !rc; ++j) rc = enqueue(c, width, height, count, descs, map, sums);
            if (!rc) { const hipError_t e = hipStreamSynchronize(c->stream); if (e != hipSuccess) { (void)hipGetLastError(); rc = map_hip_error(e); } }
            if (!rc) rc = drain_profile(c);
            if (!rc && c->prof_launches) { try { samples[k].push_back(c->prof_ms / (double)c->prof_launches); } catch (...) { rc = ENOMEM; } }
        }
    }
    (void)hipStreamSynchronize(c->stream);
    c->strip_rows = saved_rows; c->variant = saved_variant; c->profiling = saved_prof;
    c->prof_launches = saved_launches; c->prof_ms = saved_ms;
    if (rc) return rc;
    size_t best = 0;
    for (size_t k = 0; k < cand.size(); ++k) {
        if (samples[k].empty()) return ECHILD;
        std::sort(samples[k].begin(), samples[k].end());
        cand[k].ms = samples[k][samples[k].size() / 2];
        if (cand[k].ms < cand[best].ms) best = k;
    }
    // a winner must beat the default by more than the measurement's own scatter (0.5 %) to replace it
    if (best != 0 && cand[best].ms > cand[0].ms * 0.995) best = 0;
    if (best != 0) {
        try { const rmgr_ssim_hip_Context_::Tuned t = {width, height, count, mode, map, cand[best].variant, cand[best].rows}; c->tuned.push_back(t); }
        catch (...) { return ENOMEM; }
    }
    if (result) {
        rmgr_ssim_hip_TuneResult full;
        memset(&full, 0, sizeof(full));
        full.structSize = result->structSize;
        full.candidates = (rmgr_uint32_t)cand.size();
        full.bestVariant = cand[best].variant;
        full.bestStripRows = (rmgr_uint32_t)cand[best].rows;
        full.defaultMs = cand[0].ms;
        full.bestMs = cand[best].ms;
        for (size_t k = 0; k < cand.size() && k < RMGR_SSIM_HIP_TUNE_MAX_CANDIDATES; ++k) {
            full.candidateVariant[k] = cand[k].variant; full.candidateStripRows[k] = (rmgr_uint32_t)cand[k].rows; full.candidateMs[k] = cand[k].ms;
        }
        memcpy(result, &full, std::min<size_t>(result->structSize, sizeof(full)));
    }
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_clear_tuned(rmgr_ssim_hip_Context* c) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    c->tuned.clear();
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_get_tuned(const rmgr_ssim_hip_Context* c, rmgr_uint32_t index, rmgr_ssim_hip_TunedEntry* entry) RMGR_NOEXCEPT
{
    if (!c || !entry) return EINVAL;
    if (index >= c->tuned.size()) return ENOENT;
    const rmgr_ssim_hip_Context_::Tuned& t = c->tuned[index];
    entry->width = t.width; entry->height = t.height; entry->count = t.count;
    entry->withMap = t.map ? 1 : 0; entry->mode = t.mode; entry->variant = t.variant; entry->stripRows = (rmgr_uint32_t)t.strip_rows;
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_set_tuned(rmgr_ssim_hip_Context* c, rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint32_t count, rmgr_int32_t withMap,
                                     rmgr_int32_t variant, rmgr_uint32_t stripRows) RMGR_NOEXCEPT
{
    if (!c || width == 0 || height == 0 || count == 0 || variant < 0 || stripRows > 0x7FFFFFFFu || (variant == 0 && stripRows == 0)) return EINVAL;
    const bool map = withMap != 0;
    const rmgr_ssim_hip_Context_::Tuned t = {width, height, count, c->mode, map, variant, (int)stripRows};
    for (size_t i = 0; i < c->tuned.size(); ++i)
        if (c->tuned[i].width == width && c->tuned[i].height == height && c->tuned[i].count == count && c->tuned[i].mode == c->mode && c->tuned[i].map == map) { c->tuned[i] = t; return 0; }
    try { c->tuned.push_back(t); } catch (...) { return ENOMEM; }
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_probe_valu(rmgr_ssim_hip_Context* c, rmgr_int32_t wavesPerSimd, rmgr_int32_t streamKind, rmgr_int32_t launches, double* teraLaneOps,
                                      double* shaderMHz, double* slowestXcdMHz) RMGR_NOEXCEPT
{
    if (!c || !teraLaneOps || launches < 1 || launches > 64 || (streamKind != 0 && streamKind != 1)) return EINVAL;
    if (wavesPerSimd != 1 && wavesPerSimd != 2 && wavesPerSimd != 3 && wavesPerSimd != 4 && wavesPerSimd != 8) return EINVAL;
    USE_DEVICE(c);
    // ~2 ms per launch at any occupancy (the strip kernel's own duration on the headline batch): a SIMD retires one packed instruction per
    // 4.2 ... 4.9 clocks, so W waves x 24 instructions x iters / 2.4 GHz ~ 2 ms  ->  iters ~ 40000 / W
    const int iters = 40000 / wavesPerSimd;
    int rc = grow_device(c->partials, c->partials_cap, 64 + ssim_hip::kClockWords);        // the kernel's (never written) output pointer + the clock counters
    if (rc) return rc;
    uint64_t* clock = NULL;                                        // the timed launches' first workgroups (one per XCD) report the shader clock they ran at
    if (shaderMHz || slowestXcdMHz) {
        clock = reinterpret_cast<uint64_t*>(c->partials + 64);
        HIP_TRY(hipMemsetAsync(clock, 0, ssim_hip::kClockWords * sizeof(uint64_t), c->stream));
    }
    // A BURST is enqueued back to back -- untimed launches (20 in the first burst, ~40 ms: the chip takes ~25 ms of sustained load to leave its idle clock, and every
    // host-side wait between launches is an idle gap after which it ramps again), then the timed ones, each between two events -- and waited for once.  THREE bursts, the
    // best one counts: a burst sometimes runs in a degraded mode for its whole length -- at an unchanged shader clock on every XCD the rate is what W - 1 concurrent
    // waves followed by a lone one would give (two waves: 51 T, the ONE-wave rate, instead of 65...68; three: 58 / 70; four: 62 / 71; eight: 68 / 72) -- about one burst
    // in four, more often right after short or sparse launches, never two calls alike (profiles/r06_probe_bimodal.txt; the strip kernels show nothing of the kind).  The
    // yardstick is what the device CAN sustain: the best burst's median.
    std::vector<hipEvent_t> ev;
    hipError_t err = hipSuccess;
    try { ev.assign((size_t)launches + 1, (hipEvent_t)NULL); } catch (...) { return ENOMEM; }
    for (size_t k = 0; k < ev.size() && err == hipSuccess; ++k) err = hipEventCreate(&ev[k]);
    float best = 0.f;
    uint64_t best_clock[ssim_hip::kClockWords];
    memset(best_clock, 0, sizeof(best_clock));
    for (int burst = 0; burst < 3 && err == hipSuccess; ++burst) {
        if (clock) err = hipMemsetAsync(clock, 0, ssim_hip::kClockWords * sizeof(uint64_t), c->stream);
        for (int k = 0; k < (burst == 0 ? 20 : 6) && err == hipSuccess; ++k)
            err = ssim_hip::launch_probe_valu(wavesPerSimd, streamKind, c->cu_count, c->xcd_count, iters, reinterpret_cast<float*>(c->partials), c->stream, NULL);
        if (err == hipSuccess) err = hipEventRecord(ev[0], c->stream);
        for (int k = 0; k < launches && err == hipSuccess; ++k) {
            err = ssim_hip::launch_probe_valu(wavesPerSimd, streamKind, c->cu_count, c->xcd_count, iters, reinterpret_cast<float*>(c->partials), c->stream, clock);
            if (err == hipSuccess) err = hipEventRecord(ev[k + 1], c->stream);
        }
        if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
        float ms[64];
        for (int k = 0; k < launches && err == hipSuccess; ++k) err = hipEventElapsedTime(&ms[k], ev[k], ev[k + 1]);
        if (err != hipSuccess) break;
        std::sort(ms, ms + launches);
        const float med = ms[launches / 2];
        if (med > 0.f && (best == 0.f || med < best)) {
            best = med;
            if (clock) err = hipMemcpy(best_clock, clock, sizeof(best_clock), hipMemcpyDeviceToHost);
        }
    }
    for (size_t k = 0; k < ev.size(); ++k) if (ev[k]) (void)hipEventDestroy(ev[k]);
    if (err != hipSuccess) { (void)hipGetLastError(); return map_hip_error(err); }
    if (!(best > 0.f)) return ECHILD;
    *teraLaneOps = (double)ssim_hip::probe_valu_lane_ops(wavesPerSimd, c->cu_count, iters) / ((double)best * 1e-3) / 1e12;
    if (clock) clocks_from(best_clock, c->xcd_count, c->wall_clock_khz, shaderMHz, slowestXcdMHz, NULL);
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_synth_pair_device(rmgr_ssim_hip_Context* c, rmgr_uint8_t* imgA, ptrdiff_t strideA, rmgr_uint8_t* imgB, ptrdiff_t strideB,
                                             rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint64_t seed) RMGR_NOEXCEPT
{
    if (!c || !imgA || !imgB) return EINVAL;
    USE_DEVICE(c);
    HIP_TRY(ssim_hip::launch_synth_pair(imgA, strideA, imgB, strideB, width, height, seed, c->stream));
    return 0;
}

const char* rmgr_ssim_hip_describe(rmgr_ssim_hip_Context* c) RMGR_NOEXCEPT
{
    return c ? c->describe : "rmgr-ssim hip backend (no context)";
}

} // extern "C"
