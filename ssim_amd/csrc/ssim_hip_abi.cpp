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
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

using ssim_hip::PairDesc;

struct rmgr_ssim_hip_Context_ {
    int         device;
    int         cu_count;
    hipStream_t stream;
    bool        owns_stream;
    int         mode;
    int         strip_rows;
    int         variant;

    // grow-only device scratch
    double*   partials;     size_t partials_cap;   // doubles
    PairDesc* descs;        size_t descs_cap;      // entries
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
    uint8_t*  h_stage;      size_t h_stage_cap;    // bytes: small image pairs are gathered here for one DMA
    float*    h_map[2];     size_t h_map_cap[2];   // floats: bounce buffers for the map copy-back
    hipEvent_t map_ev[2];
    PairDesc* h_descs;      size_t h_descs_cap;
    size_t    descs_live;   // entries of `descs` that mirror h_descs (0: nothing uploaded)

    bool profiling;
    std::vector<std::pair<hipEvent_t, hipEvent_t> > pending;   // recorded, not yet read
    std::vector<std::pair<hipEvent_t, hipEvent_t> > free_events;
    uint64_t prof_launches;
    double   prof_ms;

    ncclComm_t comm;          // RCCL communicator (rmgr_ssim_hip_comm_*), NULL until comm_init

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

int record_begin(rmgr_ssim_hip_Context* c, hipEvent_t& b, hipEvent_t& e)
{
    b = e = NULL;
    if (!c->profiling) return 0;
    if (!c->free_events.empty()) {
        b = c->free_events.back().first; e = c->free_events.back().second;
        c->free_events.pop_back();
    } else {
        HIP_TRY(hipEventCreate(&b));
        HIP_TRY(hipEventCreate(&e));
    }
    try { c->pending.push_back(std::make_pair(b, e)); }
    catch (...) { (void)hipEventDestroy(b); (void)hipEventDestroy(e); b = e = NULL; return ENOMEM; }
    return 0;
}

int drain_profile(rmgr_ssim_hip_Context* c)
{
    for (size_t i = 0; i < c->pending.size(); ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(c->pending[i].second));
        HIP_TRY(hipEventElapsedTime(&ms, c->pending[i].first, c->pending[i].second));
        c->prof_ms += ms;
        c->prof_launches += 1;
        try { c->free_events.push_back(c->pending[i]); }
        catch (...) { (void)hipEventDestroy(c->pending[i].first); (void)hipEventDestroy(c->pending[i].second); }
    }
    c->pending.clear();
    return 0;
}

// Enqueue kernel + reduction for `count` pairs whose descriptors are in `descs` (host).
int enqueue(rmgr_ssim_hip_Context* c, uint32_t width, uint32_t height, uint32_t count, const PairDesc* descs, bool any_map, double* sums_dev)
{
    int variant = c->variant;
    for (uint32_t i = 0; i < count && variant != 1; ++i)
        if (!ssim_hip::fits_strip2(descs[i], width, height)) variant = 1;
    const ssim_hip::Geometry geo = ssim_hip::plan(width, height, count, c->mode, c->strip_rows, variant, c->cu_count);
    int rc = grow_device(c->partials, c->partials_cap, (size_t)count * geo.partials_per_image() + 1);
    if (rc) return rc;
    PairDesc single = descs[0];
    const PairDesc* descs_dev = NULL;
    if (count > 1) {
        // Re-enqueueing the same batch (the steady state of a serving loop) reuses the uploaded
        // descriptor table; a different batch waits for the stream before the pinned mirror and
        // the device table are overwritten.
        if (!(c->descs_live == count && memcmp(c->h_descs, descs, sizeof(PairDesc) * count) == 0)) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->descs_live = 0;
            if ((rc = grow_device(c->descs, c->descs_cap, count))) return rc;
            if ((rc = grow_pinned(c->h_descs, c->h_descs_cap, count))) return rc;
            memcpy(c->h_descs, descs, sizeof(PairDesc) * count);
            HIP_TRY(hipMemcpyAsync(c->descs, c->h_descs, sizeof(PairDesc) * count, hipMemcpyHostToDevice, c->stream));
            c->descs_live = count;
        }
        descs_dev = c->descs;
        if (any_map && !single.map) single.map = reinterpret_cast<float*>(1);  // only its non-NULLness is used
    }
    hipEvent_t eb, ee;
    if ((rc = record_begin(c, eb, ee))) return rc;
    HIP_TRY(ssim_hip::launch(geo, c->mode, variant, ssim_hip::interleaved_group(descs, count), descs_dev, single, c->partials, sums_dev, c->stream, eb, ee));
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

rmgr_ssim_hip_Context* g_default = NULL;
int                    g_default_err = 0;
std::once_flag         g_default_once;

rmgr_ssim_hip_Context* default_context(int* err)
{
    std::call_once(g_default_once, []() {
        int dev = 0;
        if (const char* s = getenv("RMGR_SSIM_HIP_DEVICE")) dev = atoi(s);
        g_default_err = rmgr_ssim_hip_create(&g_default, dev, NULL);
        if (g_default && g_default_err == 0) {
            const char* m = getenv("RMGR_SSIM_HIP_MODE");
            if (m && atoi(m) >= RMGR_SSIM_HIP_MODE_EXACT && atoi(m) <= RMGR_SSIM_HIP_MODE_UNFUSED) g_default->mode = atoi(m);
#if defined(RMGR_SSIM_USE_DOUBLE) && RMGR_SSIM_USE_DOUBLE
            else g_default->mode = RMGR_SSIM_HIP_MODE_DOUBLE;
#endif
        }
    });
    *err = g_default_err;
    return g_default;
}

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
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    rmgr_ssim_hip_Context* c = new (std::nothrow) rmgr_ssim_hip_Context_();
    if (!c) return ENOMEM;
    c->device = device;
    c->cu_count = prop.multiProcessorCount;
    c->stream = static_cast<hipStream_t>(stream);
    c->owns_stream = false;
    c->mode = RMGR_SSIM_HIP_MODE_EXACT;
    c->strip_rows = 0;
    c->variant = 0;
    c->partials = NULL; c->partials_cap = 0;
    c->descs = NULL; c->descs_cap = 0;
    c->stage_a = NULL; c->stage_a_cap = 0;
    c->stage_b = NULL; c->stage_b_cap = 0;
    c->stage_map = NULL; c->stage_map_cap = 0;
    c->h_sums = NULL; c->h_sums_cap = 0;
    c->h_stage = NULL; c->h_stage_cap = 0;
    for (int i = 0; i < 2; ++i) { c->slot_dev[i] = c->slot_pin[i] = NULL; c->slot_dev_cap[i] = c->slot_pin_cap[i] = 0; c->slot_copied[i] = c->slot_done[i] = NULL; }
    c->batch_sums = NULL; c->batch_sums_cap = 0;
    c->copy_stream = NULL;
    c->h_descs = NULL; c->h_descs_cap = 0;
    c->descs_live = 0;
    c->h_map[0] = c->h_map[1] = NULL; c->h_map_cap[0] = c->h_map_cap[1] = 0;
    c->map_ev[0] = c->map_ev[1] = NULL;
    c->comm = NULL;
    c->profiling = false;
    c->prof_launches = 0;
    c->prof_ms = 0.0;
    if (!stream) {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return map_hip_error(e); }
        c->owns_stream = true;
    }
    snprintf(c->describe, sizeof(c->describe), "%s %s, %d CUs, %.0f MHz, %.1f GiB; rmgr-ssim hip backend (code object gfx950)",
             prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000.0, prop.totalGlobalMem / 1073741824.0);
    *out = c;
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_destroy(rmgr_ssim_hip_Context* c) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)rmgr_ssim_hip_comm_destroy(c);
    for (size_t i = 0; i < c->pending.size(); ++i) { (void)hipEventDestroy(c->pending[i].first); (void)hipEventDestroy(c->pending[i].second); }
    for (size_t i = 0; i < c->free_events.size(); ++i) { (void)hipEventDestroy(c->free_events[i].first); (void)hipEventDestroy(c->free_events[i].second); }
    if (c->partials) (void)hipFree(c->partials);
    if (c->descs) (void)hipFree(c->descs);
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
    if (c->h_descs) (void)hipHostFree(c->h_descs);
    for (int i = 0; i < 2; ++i) { if (c->h_map[i]) (void)hipHostFree(c->h_map[i]); if (c->map_ev[i]) (void)hipEventDestroy(c->map_ev[i]); }
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_set_mode(rmgr_ssim_hip_Context* c, rmgr_int32_t mode) RMGR_NOEXCEPT
{
    if (!c || mode < RMGR_SSIM_HIP_MODE_EXACT || mode > RMGR_SSIM_HIP_MODE_UNFUSED) return EINVAL;
    c->mode = mode;
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_get_mode(const rmgr_ssim_hip_Context* c, rmgr_int32_t* mode) RMGR_NOEXCEPT
{
    if (!c || !mode) return EINVAL;
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

rmgr_int32_t rmgr_ssim_hip_get_plan(const rmgr_ssim_hip_Context* c, rmgr_uint32_t width, rmgr_uint32_t height, rmgr_uint32_t count, rmgr_ssim_hip_Plan* plan) RMGR_NOEXCEPT
{
    if (!plan) return EINVAL;
    const ssim_hip::Geometry geo = c ? ssim_hip::plan(width, height, count, c->mode, c->strip_rows, c->variant, c->cu_count)
                                     : ssim_hip::plan(width, height, count, RMGR_SSIM_HIP_MODE_EXACT, 0, 0, 256);
    plan->stripWidth = geo.strip_w;
    plan->stripRows = geo.strip_rows;
    plan->stripsX = geo.strips_x;
    plan->stripsY = geo.strips_y;
    plan->wavefronts = geo.strips_x * geo.strips_y * count;
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
    HIP_TRY(hipSetDevice(c->device));
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

rmgr_int32_t rmgr_ssim_hip_compute_ssim_batch_host(rmgr_ssim_hip_Context* c, rmgr_uint32_t count, const rmgr_ssim_Params* params, float* ssim) RMGR_NOEXCEPT
{
    if (count && (!params || !ssim)) return EINVAL;
    if (count == 0) return 0;
    for (uint32_t i = 0; i < count; ++i) {
        if (params[i].imgA.topLeft == NULL || params[i].imgB.topLeft == NULL || params[i].ssimMap != NULL) return EINVAL;
        if (params[i].width != params[0].width || params[i].height != params[0].height) return EINVAL;
    }
    int rc = 0;
    std::unique_lock<std::mutex> guard;
    if (!c) {
        c = default_context(&rc);
        if (rc) return rc;
        if (!c) return ENODEV;
        guard = std::unique_lock<std::mutex>(c->lock);
    }
    HIP_TRY(hipSetDevice(c->device));
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
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_compute_ssim_device(rmgr_ssim_hip_Context* c, float* ssim, const rmgr_ssim_Params* params) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    int rc = validate(ssim, params, NULL);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
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
    std::unique_lock<std::mutex> guard;
    if (!c) {
        c = default_context(&rc);
        if (rc) return rc;
        if (!c) return ENODEV;
        guard = std::unique_lock<std::mutex>(c->lock);
    }
    HIP_TRY(hipSetDevice(c->device));
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
    if (W && H) {
        // Stage the byte range each image occupies; step/stride semantics carry over unchanged.
        int64_t loA, hiA, loB, hiB;
        extent(params->imgA, W, H, loA, hiA);
        extent(params->imgB, W, H, loB, hiB);
        const size_t nA = (size_t)(hiA - loA + 1), nB = (size_t)(hiB - loB + 1);
        if (nA + nB <= kSmallStageBytes) {
            // Small images: two pageable copies cost ~12 us each in driver overhead.  Gather both byte ranges in
            // one pinned buffer (a ~4 us memcpy at this size) and send them with a single DMA.
            const size_t offB = (nA + 63) & ~(size_t)63;
            if ((rc = grow_pinned(c->h_stage, c->h_stage_cap, offB + nB))) return rc;
            if ((rc = grow_device(c->stage_a, c->stage_a_cap, offB + nB))) return rc;
            memcpy(c->h_stage, params->imgA.topLeft + loA, nA);
            memcpy(c->h_stage + offB, params->imgB.topLeft + loB, nB);
            HIP_TRY(hipMemcpyAsync(c->stage_a, c->h_stage, offB + nB, hipMemcpyHostToDevice, c->stream));
            dev.imgA.topLeft = c->stage_a - loA;
            dev.imgB.topLeft = c->stage_a + offB - loB;
        } else {
            if ((rc = grow_device(c->stage_a, c->stage_a_cap, nA))) return rc;
            if ((rc = grow_device(c->stage_b, c->stage_b_cap, nB))) return rc;
            HIP_TRY(hipMemcpyAsync(c->stage_a, params->imgA.topLeft + loA, nA, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->stage_b, params->imgB.topLeft + loB, nB, hipMemcpyHostToDevice, c->stream));
            dev.imgA.topLeft = c->stage_a - loA;
            dev.imgB.topLeft = c->stage_b - loB;
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
    if ((rc = enqueue(c, W, H, 1, &d, d.map != NULL, c->h_sums))) return rc;      // sum lands in pinned host memory

    if (params->ssimMap && W && H) {
        // Map back to the caller's (pageable) buffer: D2H into two pinned bounce buffers in row
        // chunks, the CPU scatters chunk k-1 into ssimStep/ssimStride layout while chunk k is in flight.
        const size_t rowsPerChunk = std::max<size_t>(1, (size_t(8) << 20) / (sizeof(float) * W));
        const size_t chunkFloats = rowsPerChunk * W;
        if ((rc = grow_pinned(c->h_map[0], c->h_map_cap[0], chunkFloats))) return rc;
        if ((rc = grow_pinned(c->h_map[1], c->h_map_cap[1], chunkFloats))) return rc;
        for (int i = 0; i < 2; ++i)
            if (!c->map_ev[i]) HIP_TRY(hipEventCreateWithFlags(&c->map_ev[i], hipEventDisableTiming));
        const size_t chunks = (H + rowsPerChunk - 1) / rowsPerChunk;
        for (size_t k = 0; k <= chunks; ++k) {
            if (k < chunks) {
                const size_t y0 = k * rowsPerChunk, rows = std::min(rowsPerChunk, (size_t)H - y0);
                HIP_TRY(hipMemcpyAsync(c->h_map[k & 1], c->stage_map + y0 * W, sizeof(float) * rows * W, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipEventRecord(c->map_ev[k & 1], c->stream));
            }
            if (k > 0) {
                const size_t j = k - 1, y0 = j * rowsPerChunk, rows = std::min(rowsPerChunk, (size_t)H - y0);
                HIP_TRY(hipEventSynchronize(c->map_ev[j & 1]));
                const float* src = c->h_map[j & 1];
                for (size_t y = 0; y < rows; ++y, src += W) {
                    float* row = params->ssimMap + (ptrdiff_t)(y0 + y) * params->ssimStride;
                    if (params->ssimStep == 1) memcpy(row, src, sizeof(float) * W);
                    else for (uint32_t x = 0; x < W; ++x) row[(ptrdiff_t)x * params->ssimStep] = src[x];
                }
            }
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
    } else {
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (ssim)
        *ssim = mean_of(c->h_sums[0], W, H);
    return 0;
}

} // extern "C"

namespace {

struct StagedPair {
    rmgr_ssim_hip_Context* c;
    std::unique_lock<std::mutex> guard;
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
    if (!c) {
        c = default_context(&rc);
        if (rc) return rc;
        if (!c) return ENODEV;
        sp.guard = std::unique_lock<std::mutex>(c->lock);
    }
    sp.c = c;
    HIP_TRY(hipSetDevice(c->device));
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
    HIP_TRY(hipSetDevice(c->device));
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
namespace {

struct Rccl {
    void* handle;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*CommDestroy)(ncclComm_t);
};

Rccl* rccl()
{
    static Rccl api;
    static std::once_flag once;
    std::call_once(once, []() {
        memset(&api, 0, sizeof(api));
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (size_t i = 0; i < sizeof(names) / sizeof(names[0]) && !api.handle; ++i)
            api.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!api.handle) return;
        api.GetUniqueId  = reinterpret_cast<ncclResult_t (*)(ncclUniqueId*)>(dlsym(api.handle, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>(dlsym(api.handle, "ncclCommInitRank"));
        api.AllReduce    = reinterpret_cast<ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)>(dlsym(api.handle, "ncclAllReduce"));
        api.CommDestroy  = reinterpret_cast<ncclResult_t (*)(ncclComm_t)>(dlsym(api.handle, "ncclCommDestroy"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) { dlclose(api.handle); api.handle = NULL; }
    });
    return api.handle ? &api : NULL;
}

int map_nccl(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess:          return 0;
    case ncclInvalidArgument:
    case ncclInvalidUsage:     return EINVAL;
    case ncclSystemError:
    case ncclUnhandledCudaError:
    case ncclInternalError:
    default:                   return ECHILD;
    }
}

} // namespace

extern "C" rmgr_int32_t rmgr_ssim_hip_comm_get_unique_id(unsigned char id[RMGR_SSIM_HIP_COMM_ID_BYTES]) RMGR_NOEXCEPT
{
    static_assert(sizeof(ncclUniqueId) == RMGR_SSIM_HIP_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id) return EINVAL;
    Rccl* r = rccl();
    if (!r) return ENOSYS;
    ncclUniqueId u;
    const int rc = map_nccl(r->GetUniqueId(&u));
    if (rc == 0) memcpy(id, &u, sizeof(u));
    return rc;
}

extern "C" rmgr_int32_t rmgr_ssim_hip_comm_init(rmgr_ssim_hip_Context* c, const unsigned char id[RMGR_SSIM_HIP_COMM_ID_BYTES],
                                                rmgr_int32_t rankCount, rmgr_int32_t rank) RMGR_NOEXCEPT
{
    if (!c || !id || rankCount < 1 || rank < 0 || rank >= rankCount || c->comm) return EINVAL;
    Rccl* r = rccl();
    if (!r) return ENOSYS;
    HIP_TRY(hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    return map_nccl(r->CommInitRank(&c->comm, rankCount, u, rank));
}

extern "C" rmgr_int32_t rmgr_ssim_hip_comm_allreduce_sums(rmgr_ssim_hip_Context* c, double* sumsDevice, rmgr_uint32_t count) RMGR_NOEXCEPT
{
    if (!c || !c->comm || (count && !sumsDevice)) return EINVAL;
    if (count == 0) return 0;
    Rccl* r = rccl();
    if (!r) return ENOSYS;
    HIP_TRY(hipSetDevice(c->device));
    return map_nccl(r->AllReduce(sumsDevice, sumsDevice, count, ncclFloat64, ncclSum, c->comm, c->stream));
}

extern "C" rmgr_int32_t rmgr_ssim_hip_comm_destroy(rmgr_ssim_hip_Context* c) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    if (!c->comm) return 0;
    Rccl* r = rccl();
    const int rc = r ? map_nccl(r->CommDestroy(c->comm)) : ENOSYS;
    c->comm = NULL;
    return rc;
}

extern "C" {

rmgr_int32_t rmgr_ssim_hip_malloc(rmgr_ssim_hip_Context* c, void** p, size_t size) RMGR_NOEXCEPT
{
    if (!c || !p) return EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(p, size ? size : 1));
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_free(rmgr_ssim_hip_Context* c, void* p) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipFree(p));
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_memcpy_h2d(rmgr_ssim_hip_Context* c, void* dst, const void* src, size_t size) RMGR_NOEXCEPT
{
    if (!c || (size && (!dst || !src))) return EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dst, src, size, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_memcpy_d2h(rmgr_ssim_hip_Context* c, void* dst, const void* src, size_t size) RMGR_NOEXCEPT
{
    if (!c || (size && (!dst || !src))) return EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_set_profiling(rmgr_ssim_hip_Context* c, rmgr_int32_t enabled) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    c->profiling = enabled != 0;
    return 0;
}

rmgr_int32_t rmgr_ssim_hip_get_profile(rmgr_ssim_hip_Context* c, rmgr_uint64_t* launches, double* kernelMs) RMGR_NOEXCEPT
{
    if (!c) return EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    int rc = drain_profile(c);
    if (rc) return rc;
    if (launches) *launches = c->prof_launches;
    if (kernelMs) *kernelMs = c->prof_ms;
    c->prof_launches = 0;
    c->prof_ms = 0.0;
    return 0;
}

const char* rmgr_ssim_hip_describe(rmgr_ssim_hip_Context* c) RMGR_NOEXCEPT
{
    return c ? c->describe : "rmgr-ssim hip backend (no context)";
}

} // extern "C"
