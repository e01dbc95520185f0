// ssim_internal.h -- the test-only selector of the reference (src/ssim_internal.h:41-53), for code that includes
// "ssim_internal.h" the way tests/rmgr-ssim-tests.cpp:231 does.  Not installed.
//
// The HIP backend reproduces the ARITHMETIC of the reference's implementations, so the selector keeps its meaning:
//   IMPL_FMA                          -> the FMA path's operation order (RMGR_SSIM_HIP_MODE_EXACT; bit-identical maps)
//   IMPL_GENERIC / SSE / SSE2 / AVX   -> separately rounded multiply-add (RMGR_SSIM_HIP_MODE_UNFUSED; bit-identical maps)
//   IMPL_AUTO, IMPL_HIP               -> the library default (EXACT: what AUTO picks on any FMA-capable x86)
//   IMPL_AVX512, IMPL_NEON            -> not reproduced: reported unsupported, and -- as in the reference when the desired
//                                        implementation is missing (src/ssim.cpp:889-892) -- the generic arithmetic is selected
// select_impl() returns the bit mask of supported implementations (0 when no gfx950 device is usable) and switches
// the process-wide default context the drop-in entry points use.
#ifndef SSIM_AMD_INTERNAL_H
#define SSIM_AMD_INTERNAL_H

#include <rmgr/ssim.h>

namespace rmgr { namespace ssim
{

enum Implementation
{
    IMPL_AUTO    = 0,
    IMPL_GENERIC = 1,
    IMPL_SSE     = 2,
    IMPL_SSE2    = 3,
    IMPL_AVX     = 4,
    IMPL_FMA     = 5,
    IMPL_AVX512  = 6,
    IMPL_NEON    = 7,
    IMPL_HIP     = 8   ///< this library's own name for its default arithmetic
};

unsigned select_impl(Implementation desiredImpl) RMGR_NOEXCEPT;

}} // namespace rmgr::ssim

#endif
