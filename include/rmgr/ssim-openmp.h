/*
 * rmgr/ssim-openmp.h -- "multi-threaded" entry points of romigrou/ssim, kept as drop-in symbols
 * (reference include/rmgr/ssim-openmp.h:50, :77, :98; src/ssim-openmp.c:40-47).
 *
 * In the reference these wrap compute_ssim() with an OpenMP thread pool over 256x64 tiles.  Here
 * the same parallelism is the GPU grid, so both spellings run the same HIP path; the symbol is
 * kept so that callers which link rmgr-ssim-openmp keep linking.
 */
#ifndef RMGR_SSIM_OPENMP_H
#define RMGR_SSIM_OPENMP_H
#ifndef RMGR_SSIM_OMP_H
    #define RMGR_SSIM_OMP_H      /* the reference's guard name (ssim-openmp.h:21), for code that tests it */
#endif

#include <rmgr/ssim.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same contract and return codes as rmgr_ssim_compute_ssim() with an all-cores thread pool. */
rmgr_int32_t rmgr_ssim_compute_ssim_openmp(float* ssim, const rmgr_ssim_Params* params) RMGR_NOEXCEPT;

#ifdef __cplusplus
} /* extern "C" */

namespace rmgr { namespace ssim
{

inline int32_t compute_ssim_openmp(float* ssim, const GeneralParams& params) RMGR_NOEXCEPT
{
    return ::rmgr_ssim_compute_ssim_openmp(ssim, &params);
}

RMGR_DEPRECATED_MSG("Use compute_ssim_openmp(float* ssim, const GeneralParams& params) instead")
inline float compute_ssim_openmp(const UnthreadedParams& params) RMGR_NOEXCEPT
{
    float ssim;
    const int32_t result = compute_ssim_openmp(&ssim, params);
    return (result == 0) ? ssim : float(-result);
}

}} /* namespace rmgr::ssim */
#endif /* __cplusplus */

#endif /* RMGR_SSIM_OPENMP_H */
