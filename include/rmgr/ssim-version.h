/*
 * rmgr/ssim-version.h -- version of the API this engine is a drop-in for.
 * The reference generates this file from src/ssim-version.h.in (project VERSION 2.1.0,
 * CMakeLists.txt:43); the numbers are kept so that callers' version checks
 * (tests/rmgr-ssim-tests.cpp:510-517) behave the same.
 */
#ifndef RMGR_SSIM_VERSION_H
#define RMGR_SSIM_VERSION_H

#define RMGR_SSIM_VERSION_MAJOR   (2)
#define RMGR_SSIM_VERSION_MINOR   (1)
#define RMGR_SSIM_VERSION_PATCH   (0)
#define RMGR_SSIM_VERSION_STRING  "2.1.0"

/* Identifies the backend behind the API (not present in the reference). */
#define RMGR_SSIM_BACKEND_STRING  "hip-gfx950"

#endif /* RMGR_SSIM_VERSION_H */
