#include <rmgr/ssim.h>
#include <rmgr/ssim-openmp.h>
/* Uses every portability macro rmgr/ssim.h exports (the reference's public macro surface,
 * include/rmgr/ssim.h:28-376 there) the way the reference's own callers do.  Compiled by
 * tests/test_abi_cpu.py as C89, C++98 and C++17 with -pedantic -Werror; it is valid in all three. */
RMGR_WARNING_PUSH()
RMGR_WARNING_MSVC_DISABLE(4996)
RMGR_WARNING_GCC_DISABLE("-Wunused-function")
RMGR_WARNING_CLANG_DISABLE("-Wunused-function")
static int silenced_unused_function(void) { return 1; }
RMGR_WARNING_POP()

#if RMGR_COMPILER_IS_GCC + RMGR_COMPILER_IS_CLANG + RMGR_COMPILER_IS_MSVC + RMGR_COMPILER_IS_DOXYGEN != 1
    #error exactly one compiler vendor expected here
#endif
#if RMGR_COMPILER_IS_GCC_OR_CLANG != (RMGR_COMPILER_IS_GCC || RMGR_COMPILER_IS_CLANG)
    #error RMGR_COMPILER_IS_GCC_OR_CLANG
#endif
#if !RMGR_COMPILER_VERSION_IS_AT_LEAST(RMGR_COMPILER_VERSION_MAJOR, RMGR_COMPILER_VERSION_MINOR, RMGR_COMPILER_VERSION_PATCH)
    #error a version is at least itself
#endif
#if RMGR_COMPILER_VERSION_IS_AT_LEAST(RMGR_COMPILER_VERSION_MAJOR + 1, 0, 0) || !RMGR_COMPILER_VERSION_IS_AT_LEAST(RMGR_COMPILER_VERSION_MAJOR - 1, 99, 99)
    #error version ordering
#endif
#if RMGR_COMPILER_IS_GCC && (!RMGR_COMPILER_IS_GCC_AT_LEAST(4,0,0) || RMGR_COMPILER_IS_GCC_LESS_THAN(4,0,0) || RMGR_COMPILER_IS_CLANG_AT_LEAST(1,0,0) || RMGR_COMPILER_IS_MSVC_LESS_THAN(99,0,0))
    #error vendor-qualified version predicates (gcc)
#endif
#if RMGR_COMPILER_IS_CLANG && (!RMGR_COMPILER_IS_CLANG_AT_LEAST(3,0,0) || RMGR_COMPILER_IS_CLANG_LESS_THAN(3,0,0) || RMGR_COMPILER_IS_GCC_AT_LEAST(1,0,0) || RMGR_COMPILER_IS_MSVC_AT_LEAST(1,0,0))
    #error vendor-qualified version predicates (clang)
#endif
#if RMGR_ARCH_IS_X86_ANY != (RMGR_ARCH_IS_X86_32 || RMGR_ARCH_IS_X86_64) || RMGR_ARCH_IS_ARM_ANY != (RMGR_ARCH_IS_ARM_32 || RMGR_ARCH_IS_ARM_64)
    #error arch families
#endif
#if defined(__x86_64__) && !(RMGR_ARCH_IS_X86_64 && RMGR_ARCH_IS_X86_ANY && !RMGR_ARCH_IS_X86_32 && !RMGR_ARCH_IS_ARM_ANY && RMGR_ARCH_IS_LITTLE_ENDIAN && !RMGR_ARCH_IS_BIG_ENDIAN)
    #error x86-64 detection
#endif
#if RMGR_ARCH_IS_LITTLE_ENDIAN && RMGR_ARCH_IS_BIG_ENDIAN
    #error byte order
#endif
#if RMGR_COMPILER_SUPPORTS_ARM_NEON && !RMGR_ARCH_IS_ARM_ANY
    #error NEON without ARM
#endif
#ifdef __cplusplus
    #if RMGR_CPP_VERSION != __cplusplus
        #error RMGR_CPP_VERSION
    #endif
#else
    #if RMGR_CPP_VERSION != 0
        #error RMGR_CPP_VERSION must be 0 in C
    #endif
#endif
#if RMGR_UINT8_MAX != 255
    #error RMGR_UINT8_MAX
#endif

static RMGR_FORCEINLINE int forced(int v) { return v + 1; }
static RMGR_NOINLINE int never_inlined(int v) { return v + 2; }
RMGR_DEPRECATED static int old_entry(int v);
RMGR_DEPRECATED_MSG("use something else") static int older_entry(int v);
static int old_entry(int v) { return v; }
static int older_entry(int v) { return v; }
static void quiet(void* context, rmgr_uint32_t job) RMGR_NOEXCEPT { (void)context; (void)job; }

#ifdef __cplusplus
extern "C"
#endif
int macro_surface_selftest(void);
int macro_surface_selftest(void)
{
    RMGR_ALIGNED_VAR(64, static float, buffer[16]);
    rmgr_ssim_ThreadFct fct = quiet;
    rmgr_uint8_t  u8 = RMGR_UINT8_MAX;
    rmgr_int32_t  i32 = -1;
    rmgr_uint32_t u32 = 1;
    rmgr_uint64_t u64 = 1;
    int ok = ((size_t)buffer % 64) == 0;
    fct(0, 0);
    RMGR_WARNING_PUSH()
    RMGR_WARNING_GCC_DISABLE("-Wdeprecated-declarations")
    RMGR_WARNING_CLANG_DISABLE("-Wdeprecated-declarations")
    ok += old_entry(0) + older_entry(0);
    RMGR_WARNING_POP()
    return ok + forced(0) + never_inlined(0) + silenced_unused_function() + (int)u8 + (int)i32 + (int)u32 + (int)u64 == 1 + 1 + 2 + 1 + 255 - 1 + 1 + 1 ? 0 : 1;
}
