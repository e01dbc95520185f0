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

extern "C" rmgr_int32_t rmgr_ssim_compute_ssim_openmp(float* ssim, const rmgr_ssim_Params* params) RMGR_NOEXCEPT
{
    // The reference builds an all-cores thread pool here; the GPU grid plays that role.
    return rmgr_ssim_hip_compute_ssim_host(NULL, ssim, params, NULL);
}

namespace rmgr { namespace ssim
{

// Values of the reference's test-only selector (src/ssim_internal.h:41-51), extended by IMPL_HIP.
enum Implementation
{
    IMPL_AUTO = 0, IMPL_GENERIC = 1, IMPL_SSE = 2, IMPL_SSE2 = 3, IMPL_AVX = 4, IMPL_FMA = 5, IMPL_AVX512 = 6, IMPL_NEON = 7,
    IMPL_HIP = 8
};

// Reports which implementations exist as a bit mask.  Only AUTO and HIP do: there is no CPU
// arithmetic in this library, so asking for a CPU ISA reports it as unsupported, which is how the
// reference's tests skip an ISA the machine lacks (tests/rmgr-ssim-tests.cpp:231-232).
unsigned select_impl(Implementation) RMGR_NOEXCEPT
{
    rmgr_int32_t devices = 0;
    rmgr_ssim_hip_get_device_count(&devices);
    return (devices > 0) ? ((1u << IMPL_AUTO) | (1u << IMPL_HIP)) : 0u;
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
