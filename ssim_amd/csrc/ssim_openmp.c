/* ssim_openmp.c -- rmgr_ssim_compute_ssim_openmp(), the one function of librmgr-ssim-openmp (reference: src/ssim-openmp.c:40-47,
 * its own static library: CMakeLists.txt:229).
 *
 * The reference builds an all-cores OpenMP thread pool here and hands it to rmgr_ssim_compute_ssim(), whose tile jobs it then
 * dispatches.  In this library the GPU grid plays that role whatever the caller passes, so the function forwards with no pool;
 * it needs no OpenMP runtime and nothing but the public API.  Like the reference's it lives in an archive of its own
 * (-lrmgr-ssim-openmp -lrmgr-ssim) and, for dynamic linking, in librmgr-ssim-hip.so next to everything else.  C89. */
#include <rmgr/ssim-openmp.h>

rmgr_int32_t rmgr_ssim_compute_ssim_openmp(float* ssim, const rmgr_ssim_Params* params) RMGR_NOEXCEPT
{
    return rmgr_ssim_compute_ssim(ssim, params, (const rmgr_ssim_ThreadPool*)0);
}
