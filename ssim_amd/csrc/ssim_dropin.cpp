// ssim_dropin.cpp -- the reference's entry points (include/rmgr/ssim.h, ssim-openmp.h) on top of
// the C ABI of <rmgr/ssim-hip.h>.  Deliberately C++98 and free of any HIP header: this is the
// "host stays C++98-visible" side of the boundary; it is compiled with -std=c++98.
//
// Reference counterparts: init helpers src/ssim.cpp:156-217, compute_ssim :933 and :1109-1120,
// select_impl :808-896, version :1126-1142, C entry :1145-1154, OpenMP adapter src/ssim-openmp.c:40-47.
#include <rmgr/ssim.h>
#include <rmgr/ssim-openmp.h>
#include <rmgr/ssim-hip.h>
#include <rmgr/ssim-version.h>
#include "ssim_internal.h"

#include <errno.h>
#include <stdlib.h>

namespace
{
    void* aligned_malloc(size_t size, size_t alignment) RMGR_NOEXCEPT
    {
        void* p = NULL;
        if (alignment < sizeof(void*))
            alignment = sizeof(void*);
        return (::posix_memalign(&p, alignment, size) == 0) ? p : NULL;
    }

    void aligned_free(void* p) RMGR_NOEXCEPT
    {
        ::free(p);
    }

    const char g_version[] = RMGR_SSIM_VERSION_STRING;
}

extern "C" rmgr_int32_t rmgr_ssim_init_interleaved(rmgr_ssim_ImgParams* params, const rmgr_uint8_t* data, ptrdiff_t imgStride,
                                                   rmgr_uint32_t channelCount, rmgr_uint32_t channelNum) RMGR_NOEXCEPT
{
    if (params == NULL || data == NULL || channelNum >= channelCount)
        return EINVAL;
    params->topLeft = data + channelNum;
    params->step    = ptrdiff_t(channelCount);
    params->stride  = imgStride;
    return 0;
}

extern "C" rmgr_int32_t rmgr_ssim_init_planar(rmgr_ssim_ImgParams* params, rmgr_uint8_t const* const planes[], const ptrdiff_t strides[],
                                              rmgr_uint32_t planeNum) RMGR_NOEXCEPT
{
    if (params == NULL || planes == NULL || planes[planeNum] == NULL || strides == NULL)
        return EINVAL;
    params->topLeft = planes[planeNum];
    params->step    = 1;
    params->stride  = strides[planeNum];
    return 0;
}

extern "C" rmgr_int32_t rmgr_ssim_use_default_allocator(rmgr_ssim_Params* params) RMGR_NOEXCEPT
{
    if (params == NULL)
        return EINVAL;
    params->alloc   = aligned_malloc;
    params->dealloc = aligned_free;
    return 0;
}

extern "C" rmgr_int32_t rmgr_ssim_get_version(rmgr_ssim_Version* version) RMGR_NOEXCEPT
{
    if (version == NULL)
        return EINVAL;
    version->major  = RMGR_SSIM_VERSION_MAJOR;
    version->minor  = RMGR_SSIM_VERSION_MINOR;
    version->patch  = RMGR_SSIM_VERSION_PATCH;
    version->string = g_version;
    return 0;
}

extern "C" rmgr_int32_t rmgr_ssim_compute_ssim(float* ssim, const rmgr_ssim_Params* params, const rmgr_ssim_ThreadPool* threadPool) RMGR_NOEXCEPT
{
    return rmgr_ssim_hip_compute_ssim_host(NULL, ssim, params, threadPool);
}

// rmgr_ssim_compute_ssim_openmp(): ssim_openmp.c (an archive of its own, like the reference's src/ssim-openmp.c)

namespace rmgr { namespace ssim
{

// The reference's test-only selector (src/ssim.cpp:808-896); semantics in ssim_internal.h.
unsigned select_impl(Implementation desiredImpl) RMGR_NOEXCEPT
{
    const unsigned supported = (1u << IMPL_AUTO) | (1u << IMPL_GENERIC) | (1u << IMPL_SSE) | (1u << IMPL_SSE2) |
                               (1u << IMPL_AVX) | (1u << IMPL_FMA) | (1u << IMPL_HIP);
    rmgr_int32_t mode;
    switch (desiredImpl) {
    case IMPL_AUTO: case IMPL_FMA: case IMPL_HIP: mode = RMGR_SSIM_HIP_MODE_EXACT;   break;
    default:                                      mode = RMGR_SSIM_HIP_MODE_UNFUSED; break;   // incl. the unsupported ones: generic arithmetic
    }
    // NULL context = the process-wide default one the drop-in calls run on; fails when no device is usable
    return (rmgr_ssim_hip_set_mode(NULL, mode) == 0) ? supported : 0u;
}

int32_t compute_ssim(float* ssim, const GeneralParams& params, const ThreadPool* threadPool) RMGR_NOEXCEPT
{
    return rmgr_ssim_hip_compute_ssim_host(NULL, ssim, &params, threadPool);
}

float compute_ssim(const Params& params) RMGR_NOEXCEPT
{
    ThreadPool pool;
    pool.dispatch    = params.threadPool;
    pool.context     = params.threadPoolContext;
    pool.threadCount = params.threadCount;

    float value;
    const int32_t rc = compute_ssim(&value, params, &pool);
    return (rc == 0) ? value : float(-rc);
}

}} // namespace rmgr::ssim
