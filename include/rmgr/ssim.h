/*
 * rmgr/ssim.h -- public API of the MI355X-native SSIM engine.
 *
 * Drop-in for the header of the same name in romigrou/ssim 2.1.0: every type, field order and
 * function signature below matches the reference's include/rmgr/ssim.h (C API :438-605,
 * C++ API :620-728) so that code written against the reference compiles and links unchanged.
 * The implementation behind it is different: the entry points forward through the C ABI of
 * <rmgr/ssim-hip.h> into HIP kernels for gfx950.  This header is plain C89 / C++98 and pulls
 * in no HIP or ROCm header.
 */
#ifndef RMGR_SSIM_H
#define RMGR_SSIM_H

#include <stddef.h>

#if defined(__cplusplus) && __cplusplus >= 201103L
    #include <cstdint>
    #define RMGR_NOEXCEPT          noexcept
#else
    #include <stdint.h>
    #ifdef __cplusplus
        #define RMGR_NOEXCEPT      throw()
    #else
        #define RMGR_NOEXCEPT
    #endif
#endif
#if defined(__cplusplus) && __cplusplus >= 201703L
    #define RMGR_NOEXCEPT_TYPEDEF  noexcept   /* noexcept is part of the function type since C++17 */
#else
    #define RMGR_NOEXCEPT_TYPEDEF
#endif

#if defined(__GNUC__) || defined(__clang__)
    #define RMGR_DEPRECATED_MSG(msg)  __attribute__((deprecated(msg)))
#else
    #define RMGR_DEPRECATED_MSG(msg)
#endif

typedef uint8_t  rmgr_uint8_t;
typedef int32_t  rmgr_int32_t;
typedef uint32_t rmgr_uint32_t;
typedef uint64_t rmgr_uint64_t;


/* ------------------------------------------------------------------------------------------- */
/* C API                                                                                       */

#ifdef __cplusplus
extern "C" {
#endif

/* Allocation callbacks (reference ssim.h:438-439). */
typedef void* (*rmgr_ssim_AllocFct)(size_t size, size_t alignment) RMGR_NOEXCEPT_TYPEDEF;
typedef void  (*rmgr_ssim_DeallocFct)(void* address) RMGR_NOEXCEPT_TYPEDEF;

/* Thread-pool callbacks (reference ssim.h:448, :466).  The HIP engine validates them exactly
 * like the reference does but never dispatches CPU jobs: the GPU grid replaces the tile jobs. */
typedef void         (*rmgr_ssim_ThreadFct)(void* arg, rmgr_uint32_t jobNum) RMGR_NOEXCEPT_TYPEDEF;
typedef rmgr_int32_t (*rmgr_ssim_ThreadPoolFct)(void* context, rmgr_ssim_ThreadFct fct, void* const args[],
                                                rmgr_uint32_t threadCount, rmgr_uint32_t jobCount) RMGR_NOEXCEPT_TYPEDEF;

typedef struct rmgr_ssim_Version_
{
    rmgr_uint32_t major;
    rmgr_uint32_t minor;
    rmgr_uint32_t patch;
    const char*   string;
} rmgr_ssim_Version;

/* One channel of one image: pixel (x,y) lives at topLeft + x*step + y*stride (bytes; either
 * distance may be negative: bottom-up, column-major, interleaved, ... all fit). */
typedef struct rmgr_ssim_ImgParams_
{
    const rmgr_uint8_t* topLeft;
    ptrdiff_t           step;
    ptrdiff_t           stride;

#ifdef __cplusplus
    rmgr_int32_t init_interleaved(const rmgr_uint8_t* data, ptrdiff_t imgStride, rmgr_uint32_t channelCount, rmgr_uint32_t channelNum) RMGR_NOEXCEPT;
    rmgr_int32_t init_planar(rmgr_uint8_t const* const planes[], const ptrdiff_t strides[], rmgr_uint32_t planeNum) RMGR_NOEXCEPT;
#endif
} rmgr_ssim_ImgParams;

/* Everything but threading.  ssimStep / ssimStride count floats, not bytes. */
typedef struct rmgr_ssim_Params_
{
    rmgr_uint32_t        width;
    rmgr_uint32_t        height;
    rmgr_ssim_ImgParams  imgA;
    rmgr_ssim_ImgParams  imgB;

    float*               ssimMap;     /* NULL: no per-pixel map wanted */
    ptrdiff_t            ssimStep;
    ptrdiff_t            ssimStride;

    rmgr_ssim_AllocFct   alloc;       /* NULL: the library manages its own staging memory */
    rmgr_ssim_DeallocFct dealloc;

#ifdef __cplusplus
    void use_default_allocator() RMGR_NOEXCEPT;
#endif
} rmgr_ssim_Params;

typedef struct rmgr_ssim_ThreadPool_
{
    rmgr_ssim_ThreadPoolFct dispatch;
    void*                   context;
    rmgr_uint32_t           threadCount;
} rmgr_ssim_ThreadPool;

/* 0, or EINVAL when version is NULL. */
rmgr_int32_t rmgr_ssim_get_version(rmgr_ssim_Version* version) RMGR_NOEXCEPT;

/* topLeft = data + channelNum, step = channelCount, stride = imgStride.  EINVAL on NULL or channelNum >= channelCount. */
rmgr_int32_t rmgr_ssim_init_interleaved(rmgr_ssim_ImgParams* params, const rmgr_uint8_t* data, ptrdiff_t imgStride, rmgr_uint32_t channelCount, rmgr_uint32_t channelNum) RMGR_NOEXCEPT;

/* topLeft = planes[planeNum], step = 1, stride = strides[planeNum]. */
rmgr_int32_t rmgr_ssim_init_planar(rmgr_ssim_ImgParams* params, rmgr_uint8_t const* const planes[], const ptrdiff_t strides[], rmgr_uint32_t planeNum) RMGR_NOEXCEPT;

/* alloc/dealloc = aligned malloc / free. */
rmgr_int32_t rmgr_ssim_use_default_allocator(rmgr_ssim_Params* params) RMGR_NOEXCEPT;

/*
 * SSIM of one channel of two 8-bit images (host pointers), global value and/or per-pixel map.
 * Returns 0, EINVAL (NULL params / both outputs NULL / NULL image / dispatch set with
 * threadCount 0), ENOMEM (params->alloc returned NULL, or device memory exhausted), ECHILD
 * (a HIP call failed) or ENODEV (no usable gfx950 device; extension to the reference's set).
 */
rmgr_int32_t rmgr_ssim_compute_ssim(float* ssim, const rmgr_ssim_Params* params, const rmgr_ssim_ThreadPool* threadPool) RMGR_NOEXCEPT;

#ifdef __cplusplus
} /* extern "C" */
#endif


/* ------------------------------------------------------------------------------------------- */
/* C++ API                                                                                     */

#ifdef __cplusplus

inline rmgr_int32_t rmgr_ssim_ImgParams::init_interleaved(const rmgr_uint8_t* data, ptrdiff_t imgStride, rmgr_uint32_t channelCount, rmgr_uint32_t channelNum) RMGR_NOEXCEPT
{
    return ::rmgr_ssim_init_interleaved(this, data, imgStride, channelCount, channelNum);
}

inline rmgr_int32_t rmgr_ssim_ImgParams::init_planar(rmgr_uint8_t const* const planes[], const ptrdiff_t strides[], rmgr_uint32_t planeNum) RMGR_NOEXCEPT
{
    return ::rmgr_ssim_init_planar(this, planes, strides, planeNum);
}

inline void rmgr_ssim_Params::use_default_allocator() RMGR_NOEXCEPT
{
    ::rmgr_ssim_use_default_allocator(this);
}

namespace rmgr { namespace ssim
{

typedef ::rmgr_uint8_t             uint8_t;
typedef ::rmgr_int32_t             int32_t;
typedef ::rmgr_uint32_t            uint32_t;
typedef ::rmgr_uint64_t            uint64_t;
typedef ::rmgr_ssim_AllocFct       AllocFct;
typedef ::rmgr_ssim_DeallocFct     DeallocFct;
typedef ::rmgr_ssim_ThreadFct      ThreadFct;
typedef ::rmgr_ssim_ThreadPoolFct  ThreadPoolFct;
typedef ::rmgr_ssim_Version        Version;
typedef ::rmgr_ssim_ImgParams      ImgParams;
typedef ::rmgr_ssim_Params         GeneralParams;
typedef ::rmgr_ssim_ThreadPool     ThreadPool;
typedef GeneralParams              UnthreadedParams;

inline Version get_version() RMGR_NOEXCEPT
{
    Version v;
    ::rmgr_ssim_get_version(&v);
    return v;
}

/* Same contract as rmgr_ssim_compute_ssim(). */
int32_t compute_ssim(float* ssim, const GeneralParams& params, const ThreadPool* threadPool=NULL) RMGR_NOEXCEPT;

/* Deprecated all-in-one parameter block, kept because the reference's tests still use it. */
struct Params: public rmgr_ssim_Params_
{
    ThreadPoolFct  threadPool;
    void*          threadPoolContext;
    uint32_t       threadCount;
};

/* Returns the SSIM, or -errno as a float on failure (see get_errno). */
RMGR_DEPRECATED_MSG("Use compute_ssim(float* ssim, const GeneralParams& params, const ThreadPool* threadPool) instead")
float compute_ssim(const Params& params) RMGR_NOEXCEPT;

RMGR_DEPRECATED_MSG("You don't need this if you use compute_ssim(float* ssim, const GeneralParams& params, const ThreadPool* threadPool)")
inline int32_t get_errno(float ssim) RMGR_NOEXCEPT
{
    return (ssim>=0) ? 0 : -int32_t(ssim);
}

}} /* namespace rmgr::ssim */

#endif /* __cplusplus */

#endif /* RMGR_SSIM_H */
