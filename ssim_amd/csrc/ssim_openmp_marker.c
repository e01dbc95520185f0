/* ssim_openmp_marker.c -- the only member of librmgr-ssim-openmp.a.
 *
 * The reference ships rmgr_ssim_compute_ssim_openmp() in a second static library built from src/ssim-openmp.c
 * (CMakeLists.txt:229).  In this library that entry point needs no OpenMP -- the GPU is the thread pool -- and is
 * defined next to the other drop-in symbols (ssim_dropin.cpp, in librmgr-ssim.a / librmgr-ssim-hip.so).  The archive
 * is kept so that "-lrmgr-ssim-openmp -lrmgr-ssim" links as it does against the reference.  C89. */
const char rmgr_ssim_openmp_archive_note[] = "rmgr_ssim_compute_ssim_openmp lives in librmgr-ssim (HIP backend)";
