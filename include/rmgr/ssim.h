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

/* =========================================================================================== */
/* Portability macros.                                                                         */
/*                                                                                             */
/* The reference's header exports its compiler / architecture / attribute helpers as part of   */
/* its public surface (reference ssim.h:28-376) and its own callers use them (CLI, sample,     */
/* tests, tests/ssim_naive.h).  They are provided here under the same names with the same      */
/* values.  As in the reference every one of them may be pre-defined by the includer: each     */
/* definition below only happens when the name is still free.                                  */
/* =========================================================================================== */

/* ---- language level ------------------------------------------------------------------------ */

#ifndef RMGR_CPP_VERSION                 /* 0 in C, else the C++ standard's date code (MSVC reports it in _MSVC_LANG) */
    #if defined(_MSVC_LANG)
        #define RMGR_CPP_VERSION  _MSVC_LANG
    #elif defined(__cplusplus)
        #define RMGR_CPP_VERSION  __cplusplus
    #else
        #define RMGR_CPP_VERSION  0
    #endif
#endif

#ifndef RMGR_NOEXCEPT                    /* C++11 on: noexcept; before, and in C: nothing */
    #if RMGR_CPP_VERSION >= 201103
        #define RMGR_NOEXCEPT  noexcept
    #else
        #define RMGR_NOEXCEPT
    #endif
#endif

#ifndef RMGR_NOEXCEPT_TYPEDEF            /* noexcept belongs to the function TYPE only since C++17 */
    #if RMGR_CPP_VERSION >= 201703
        #define RMGR_NOEXCEPT_TYPEDEF  RMGR_NOEXCEPT
    #else
        #define RMGR_NOEXCEPT_TYPEDEF
    #endif
#endif

/* ---- which compiler -------------------------------------------------------------------------
 * One vendor is picked, in this precedence: Doxygen (when the includer did not say
 * RMGR_COMPILER_IS_NOT_DOXYGEN -- this header says it by default), Clang (which also poses as
 * GCC and, as clang-cl, as MSVC), MSVC, GCC.  The private selector below is undefined again
 * at the end of the section. */

#ifndef RMGR_COMPILER_IS_NOT_DOXYGEN
    #define RMGR_COMPILER_IS_NOT_DOXYGEN  1
#endif

#if !defined(RMGR_COMPILER_IS_NOT_DOXYGEN)
    #define RMGR_SSIM_PRIV_VENDOR  4
#elif defined(__clang__)
    #define RMGR_SSIM_PRIV_VENDOR  1
#elif defined(_MSC_VER)
    #define RMGR_SSIM_PRIV_VENDOR  2
#elif defined(__GNUC__)
    #define RMGR_SSIM_PRIV_VENDOR  3
#else
    #define RMGR_SSIM_PRIV_VENDOR  0
#endif

#ifndef RMGR_COMPILER_IS_DOXYGEN
    #if RMGR_SSIM_PRIV_VENDOR == 4
        #define RMGR_COMPILER_IS_DOXYGEN  1
    #else
        #define RMGR_COMPILER_IS_DOXYGEN  0
    #endif
#endif
#ifndef RMGR_COMPILER_IS_CLANG
    #if RMGR_SSIM_PRIV_VENDOR == 1
        #define RMGR_COMPILER_IS_CLANG  1
    #else
        #define RMGR_COMPILER_IS_CLANG  0
    #endif
#endif
#ifndef RMGR_COMPILER_IS_MSVC
    #if RMGR_SSIM_PRIV_VENDOR == 2
        #define RMGR_COMPILER_IS_MSVC  1
    #else
        #define RMGR_COMPILER_IS_MSVC  0
    #endif
#endif
#ifndef RMGR_COMPILER_IS_GCC
    #if RMGR_SSIM_PRIV_VENDOR == 3
        #define RMGR_COMPILER_IS_GCC  1
    #else
        #define RMGR_COMPILER_IS_GCC  0
    #endif
#endif
#ifndef RMGR_COMPILER_IS_GCC_OR_CLANG
    #define RMGR_COMPILER_IS_GCC_OR_CLANG  (RMGR_COMPILER_IS_GCC || RMGR_COMPILER_IS_CLANG)
#endif

/* The vendor's version triple (only defined when the vendor is known, as in the reference). */
#if RMGR_SSIM_PRIV_VENDOR == 1
    #ifndef RMGR_COMPILER_VERSION_MAJOR
        #define RMGR_COMPILER_VERSION_MAJOR  __clang_major__
    #endif
    #ifndef RMGR_COMPILER_VERSION_MINOR
        #define RMGR_COMPILER_VERSION_MINOR  __clang_minor__
    #endif
    #ifndef RMGR_COMPILER_VERSION_PATCH
        #define RMGR_COMPILER_VERSION_PATCH  __clang_patchlevel__
    #endif
#elif RMGR_SSIM_PRIV_VENDOR == 2         /* _MSC_VER = MMmm, _MSC_FULL_VER = MMmmppppp */
    #ifndef RMGR_COMPILER_VERSION_MAJOR
        #define RMGR_COMPILER_VERSION_MAJOR  (_MSC_VER / 100)
    #endif
    #ifndef RMGR_COMPILER_VERSION_MINOR
        #define RMGR_COMPILER_VERSION_MINOR  (_MSC_VER % 100)
    #endif
    #ifndef RMGR_COMPILER_VERSION_PATCH
        #define RMGR_COMPILER_VERSION_PATCH  (_MSC_FULL_VER % 100000)
    #endif
#elif RMGR_SSIM_PRIV_VENDOR == 3
    #ifndef RMGR_COMPILER_VERSION_MAJOR
        #define RMGR_COMPILER_VERSION_MAJOR  __GNUC__
    #endif
    #ifndef RMGR_COMPILER_VERSION_MINOR
        #define RMGR_COMPILER_VERSION_MINOR  __GNUC_MINOR__
    #endif
    #ifndef RMGR_COMPILER_VERSION_PATCH
        #define RMGR_COMPILER_VERSION_PATCH  __GNUC_PATCHLEVEL__
    #endif
#endif

/* (major, minor, patch) <= the compiler's own triple, compared lexicographically. */
#ifndef RMGR_COMPILER_VERSION_IS_AT_LEAST
    #define RMGR_COMPILER_VERSION_IS_AT_LEAST(major,minor,patch)                           \
        (  RMGR_COMPILER_VERSION_MAJOR >  (major)                                          \
        || (RMGR_COMPILER_VERSION_MAJOR == (major) && RMGR_COMPILER_VERSION_MINOR >  (minor)) \
        || (RMGR_COMPILER_VERSION_MAJOR == (major) && RMGR_COMPILER_VERSION_MINOR == (minor) && RMGR_COMPILER_VERSION_PATCH >= (patch)))
#endif
#ifndef RMGR_COMPILER_IS_CLANG_AT_LEAST
    #define RMGR_COMPILER_IS_CLANG_AT_LEAST(major,minor,patch)   (RMGR_COMPILER_IS_CLANG && RMGR_COMPILER_VERSION_IS_AT_LEAST((major),(minor),(patch)))
#endif
#ifndef RMGR_COMPILER_IS_CLANG_LESS_THAN
    #define RMGR_COMPILER_IS_CLANG_LESS_THAN(major,minor,patch)  (RMGR_COMPILER_IS_CLANG && !RMGR_COMPILER_VERSION_IS_AT_LEAST((major),(minor),(patch)))
#endif
#ifndef RMGR_COMPILER_IS_MSVC_AT_LEAST
    #define RMGR_COMPILER_IS_MSVC_AT_LEAST(major,minor,patch)    (RMGR_COMPILER_IS_MSVC && RMGR_COMPILER_VERSION_IS_AT_LEAST((major),(minor),(patch)))
#endif
#ifndef RMGR_COMPILER_IS_MSVC_LESS_THAN
    #define RMGR_COMPILER_IS_MSVC_LESS_THAN(major,minor,patch)   (RMGR_COMPILER_IS_MSVC && !RMGR_COMPILER_VERSION_IS_AT_LEAST((major),(minor),(patch)))
#endif
#ifndef RMGR_COMPILER_IS_GCC_AT_LEAST
    #define RMGR_COMPILER_IS_GCC_AT_LEAST(major,minor,patch)     (RMGR_COMPILER_IS_GCC && RMGR_COMPILER_VERSION_IS_AT_LEAST((major),(minor),(patch)))
#endif
#ifndef RMGR_COMPILER_IS_GCC_LESS_THAN
    #define RMGR_COMPILER_IS_GCC_LESS_THAN(major,minor,patch)    (RMGR_COMPILER_IS_GCC && !RMGR_COMPILER_VERSION_IS_AT_LEAST((major),(minor),(patch)))
#endif

/* ---- diagnostics control ---------------------------------------------------------------------
 * RMGR_WARNING_PUSH() / RMGR_WARNING_POP() bracket a region; inside it
 * RMGR_WARNING_GCC_DISABLE("-Wfoo"), RMGR_WARNING_CLANG_DISABLE("-Wfoo") and
 * RMGR_WARNING_MSVC_DISABLE(4996) silence one diagnostic for the vendor they name and expand to
 * nothing for every other vendor.  The two-level spelling is needed to paste the option string
 * into the pragma text before it is stringified. */

#if RMGR_SSIM_PRIV_VENDOR == 1
    #ifndef RMGR_WARNING_PUSH
        #define RMGR_WARNING_PUSH()                    _Pragma("clang diagnostic push")
    #endif
    #ifndef RMGR_WARNING_POP
        #define RMGR_WARNING_POP()                     _Pragma("clang diagnostic pop")
    #endif
    #ifndef RMGR_WARNING_CLANG_DISABLE
        #define RMGR_WARNING_CLANG_DO_DISABLE(string)  _Pragma(#string)
        #define RMGR_WARNING_CLANG_DISABLE(name)       RMGR_WARNING_CLANG_DO_DISABLE(clang diagnostic ignored name)
    #endif
#elif RMGR_SSIM_PRIV_VENDOR == 2
    #ifndef RMGR_WARNING_PUSH
        #define RMGR_WARNING_PUSH()                    __pragma(warning(push))
    #endif
    #ifndef RMGR_WARNING_POP
        #define RMGR_WARNING_POP()                     __pragma(warning(pop))
    #endif
    #ifndef RMGR_WARNING_MSVC_DISABLE
        #define RMGR_WARNING_MSVC_DISABLE(number)      __pragma(warning(disable: number))
    #endif
#elif RMGR_SSIM_PRIV_VENDOR == 3
    #ifndef RMGR_WARNING_PUSH
        #define RMGR_WARNING_PUSH()                    _Pragma("GCC diagnostic push")
    #endif
    #ifndef RMGR_WARNING_POP
        #define RMGR_WARNING_POP()                     _Pragma("GCC diagnostic pop")
    #endif
    #ifndef RMGR_WARNING_GCC_DISABLE
        #define RMGR_WARNING_GCC_DO_DISABLE(string)    _Pragma(#string)
        #define RMGR_WARNING_GCC_DISABLE(name)         RMGR_WARNING_GCC_DO_DISABLE(GCC diagnostic ignored name)
    #endif
#endif
#ifndef RMGR_WARNING_PUSH
    #define RMGR_WARNING_PUSH()
#endif
#ifndef RMGR_WARNING_POP
    #define RMGR_WARNING_POP()
#endif
#ifndef RMGR_WARNING_GCC_DISABLE
    #define RMGR_WARNING_GCC_DISABLE(name)
#endif
#ifndef RMGR_WARNING_CLANG_DISABLE
    #define RMGR_WARNING_CLANG_DISABLE(name)
#endif
#ifndef RMGR_WARNING_MSVC_DISABLE
    #define RMGR_WARNING_MSVC_DISABLE(number)
#endif

/* ---- declaration attributes ------------------------------------------------------------------ */

#ifndef RMGR_DEPRECATED
    #if RMGR_SSIM_PRIV_VENDOR == 1 || RMGR_SSIM_PRIV_VENDOR == 3
        #define RMGR_DEPRECATED  __attribute__((deprecated))
    #elif RMGR_SSIM_PRIV_VENDOR == 2
        #define RMGR_DEPRECATED  __declspec(deprecated)
    #else
        #define RMGR_DEPRECATED
    #endif
#endif
#ifndef RMGR_DEPRECATED_MSG
    #if RMGR_SSIM_PRIV_VENDOR == 1 || RMGR_SSIM_PRIV_VENDOR == 3
        #define RMGR_DEPRECATED_MSG(msg)  __attribute__((deprecated(msg)))
    #elif RMGR_SSIM_PRIV_VENDOR == 2
        #define RMGR_DEPRECATED_MSG(msg)  __declspec(deprecated(msg))
    #else
        #define RMGR_DEPRECATED_MSG(msg)
    #endif
#endif

/* RMGR_ALIGNED_VAR(64, float, buf[16]);  declares `float buf[16]` on a 64-byte boundary. */
#ifndef RMGR_ALIGNED_VAR
    #if RMGR_SSIM_PRIV_VENDOR == 1 || RMGR_SSIM_PRIV_VENDOR == 3
        #define RMGR_ALIGNED_VAR(alignment, type, name)  type name __attribute__((aligned(alignment)))
    #elif RMGR_SSIM_PRIV_VENDOR == 2
        #define RMGR_ALIGNED_VAR(alignment, type, name)  __declspec(align(alignment)) type name
    #elif RMGR_CPP_VERSION >= 201103
        #define RMGR_ALIGNED_VAR(alignment, type, name)  alignas(alignment) type name
    #else
        #define RMGR_ALIGNED_VAR(alignment, type, name)  type name
    #endif
#endif

#ifndef RMGR_FORCEINLINE
    #if RMGR_COMPILER_IS_MSVC
        #define RMGR_FORCEINLINE  __forceinline
    #elif RMGR_COMPILER_IS_GCC_OR_CLANG
        #define RMGR_FORCEINLINE  __inline__ __attribute__((always_inline))   /* __inline__: also valid in C89 */
    #else
        #define RMGR_FORCEINLINE  inline
    #endif
#endif
#ifndef RMGR_NOINLINE
    #if RMGR_COMPILER_IS_MSVC
        #define RMGR_NOINLINE  __declspec(noinline)
    #elif RMGR_COMPILER_IS_GCC_OR_CLANG
        #define RMGR_NOINLINE  __attribute__((noinline))
    #else
        #define RMGR_NOINLINE
    #endif
#endif

#undef RMGR_SSIM_PRIV_VENDOR

/* ---- target architecture --------------------------------------------------------------------- */

#ifndef RMGR_ARCH_IS_X86_32
    #if defined(__i386__) || defined(_M_IX86)
        #define RMGR_ARCH_IS_X86_32  1
    #else
        #define RMGR_ARCH_IS_X86_32  0
    #endif
#endif
#ifndef RMGR_ARCH_IS_X86_64
    #if defined(__amd64__) || defined(_M_AMD64) || defined(_M_X64)
        #define RMGR_ARCH_IS_X86_64  1
    #else
        #define RMGR_ARCH_IS_X86_64  0
    #endif
#endif
#ifndef RMGR_ARCH_IS_X86_ANY
    #define RMGR_ARCH_IS_X86_ANY  (RMGR_ARCH_IS_X86_32 || RMGR_ARCH_IS_X86_64)
#endif
#ifndef RMGR_ARCH_IS_ARM_32
    #if defined(__arm__) || defined(_M_ARM)
        #define RMGR_ARCH_IS_ARM_32  1
    #else
        #define RMGR_ARCH_IS_ARM_32  0
    #endif
#endif
#ifndef RMGR_ARCH_IS_ARM_64
    #if defined(__aarch64__) || defined(_M_ARM64)
        #define RMGR_ARCH_IS_ARM_64  1
    #else
        #define RMGR_ARCH_IS_ARM_64  0
    #endif
#endif
#ifndef RMGR_ARCH_IS_ARM_ANY
    #define RMGR_ARCH_IS_ARM_ANY  (RMGR_ARCH_IS_ARM_32 || RMGR_ARCH_IS_ARM_64)
#endif

/* Byte order.  Either macro may be pre-defined and then decides the other; otherwise the
 * compiler's __BYTE_ORDER__ is believed, then "x86 and Windows are little endian"; a target that
 * offers none of these clues is reported as neither. */
#ifndef RMGR_ARCH_IS_LITTLE_ENDIAN
    #if defined(RMGR_ARCH_IS_BIG_ENDIAN) && RMGR_ARCH_IS_BIG_ENDIAN
        #define RMGR_ARCH_IS_LITTLE_ENDIAN  0
    #elif defined(__BYTE_ORDER__) && defined(__ORDER_LITTLE_ENDIAN__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
        #define RMGR_ARCH_IS_LITTLE_ENDIAN  1
    #elif RMGR_ARCH_IS_X86_ANY || defined(_WIN32)
        #define RMGR_ARCH_IS_LITTLE_ENDIAN  1
    #else
        #define RMGR_ARCH_IS_LITTLE_ENDIAN  0
    #endif
#endif
#ifndef RMGR_ARCH_IS_BIG_ENDIAN
    #if RMGR_ARCH_IS_LITTLE_ENDIAN
        #define RMGR_ARCH_IS_BIG_ENDIAN  0
    #elif defined(__BYTE_ORDER__) && defined(__ORDER_BIG_ENDIAN__) && __BYTE_ORDER__ == __ORDER_BIG_ENDIAN__
        #define RMGR_ARCH_IS_BIG_ENDIAN  1
    #else
        #define RMGR_ARCH_IS_BIG_ENDIAN  0
    #endif
#endif
#if RMGR_ARCH_IS_LITTLE_ENDIAN && RMGR_ARCH_IS_BIG_ENDIAN
    #error RMGR_ARCH_IS_LITTLE_ENDIAN and RMGR_ARCH_IS_BIG_ENDIAN are both set
#endif

#ifndef RMGR_COMPILER_SUPPORTS_ARM_NEON  /* AArch64 always has it (MSVC/ARM64 excepted, as upstream); 32-bit ARM when enabled */
    #if (RMGR_ARCH_IS_ARM_64 && !RMGR_COMPILER_IS_MSVC) || (RMGR_ARCH_IS_ARM_32 && (defined(__ARM_NEON) || RMGR_COMPILER_IS_MSVC))
        #define RMGR_COMPILER_SUPPORTS_ARM_NEON  1
    #else
        #define RMGR_COMPILER_SUPPORTS_ARM_NEON  0
    #endif
#endif

/* ---- fixed-width integers --------------------------------------------------------------------- */

#if RMGR_COMPILER_IS_MSVC_LESS_THAN(16,0,0)      /* before Visual C++ 2010 there is no <stdint.h> */
    typedef unsigned __int8   rmgr_uint8_t;
    typedef signed   __int32  rmgr_int32_t;
    typedef unsigned __int32  rmgr_uint32_t;
    typedef unsigned __int64  rmgr_uint64_t;
    #define RMGR_UINT8_MAX  255
#else
    #include <stdint.h>
    typedef uint8_t   rmgr_uint8_t;
    typedef int32_t   rmgr_int32_t;
    typedef uint32_t  rmgr_uint32_t;
    typedef uint64_t  rmgr_uint64_t;
    #define RMGR_UINT8_MAX  UINT8_MAX
#endif


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
