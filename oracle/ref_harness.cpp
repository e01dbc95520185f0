/*
 * ref_harness.cpp -- drives the REAL reference kernels, compiled from /root/reference where
 * they lie (never copied), so the restatement in ssim_oracle.c can be validated bit-for-bit.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as ssim_oracle.c).  Built by oracle/Makefile into
 * oracle/_ref/libssim_ref.so, which is git-ignored and travels to the GPU box only as a
 * prebuilt binary.
 *
 * What is the reference's own code here (object files of the unmodified sources):
 *   - rmgr::ssim::avx::g_multiplyFct      src/ssim_avx.cpp:76-109   (-mavx)
 *   - rmgr::ssim::fma::g_gaussianBlurFct  src/ssim_fma.cpp:106-276  (-mfma)
 *   - rmgr::ssim::avx::g_gaussianBlurFct  src/ssim_avx.cpp:115-293  (-mavx)
 *   - rmgr::ssim::avx::g_sumTileFct       src/ssim_avx.cpp:299-412  (-mavx)
 *   - rmgr::ssim::naive::retrieve_tile / compute_ssim   tests/ssim_naive.h (header-only)
 * What is NOT: the tile loop below.  src/ssim.cpp (the driver) cannot be compiled directly
 * because it includes <rmgr/ssim-version.h>, which only the reference's cmake configure step
 * generates; per the build rules that makes the driver TU unbuildable here, so this harness
 * re-creates just the driver's buffer choreography (src/ssim.cpp:747-783, :1026-1103) around
 * the real kernels.  The halo fetch uses the reference's own naive::retrieve_tile template,
 * which is the same algorithm as src/ssim.cpp:515-583.
 */
#include "ssim_internal.h"   // from /root/reference/src (via -I)
#include "ssim_naive.h"      // from /root/reference/tests (via -I)
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

typedef rmgr::ssim::Float Float;

// src/ssim.cpp:227-239
const uint32_t kRadius     = 5;
const uint32_t kTileW      = 256;
const uint32_t kTileH      = 64;
const uint32_t kVertMargin = 2 * kRadius;
const size_t   kRowAlign   = 64 / sizeof(Float);
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) & ~(a - 1); }
const size_t kBufStride   = align_up(kTileW + 2 * kRadius, kRowAlign);
const size_t kBufCapacity = align_up(kBufStride * (kTileH + 2 * kVertMargin) + kRowAlign - 1, kRowAlign);

struct Job {
    uint32_t width, height;
    const uint8_t* a; ptrdiff_t aStep, aStride;
    const uint8_t* b; ptrdiff_t bStep, bStride;
    float* map; ptrdiff_t mapStep, mapStride;
    rmgr::ssim::GaussianBlurFct blur;
};

// src/ssim.cpp:747-783 (process_tile): same six-buffer layout and aliasing, real kernels.
double run_tile(const Job& jb, Float* buffers, uint32_t tileX, uint32_t tileY)
{
    const uint32_t tw = std::min(kTileW, jb.width - tileX);
    const uint32_t th = std::min(kTileH, jb.height - tileY);
    const uint32_t ts = uint32_t(align_up(tw + 2 * kRadius, kRowAlign));
    const size_t   off = kVertMargin * ts + align_up(kRadius, kRowAlign);
    Float* buf[6];
    for (int i = 0; i < 6; ++i)
        buf[i] = buffers + i * kBufCapacity;

    Float* a = buf[1] + off;
    Float* b = buf[2] + off;
    // naive::retrieve_tile takes the pointer to the start of the margin area (tests/ssim_naive.h:155-221)
    rmgr::ssim::naive::retrieve_tile(a - (kRadius * ts + kRadius), tw, th, ts, kRadius, tileX, tileY, jb.a, jb.width, jb.height, jb.aStep, jb.aStride);
    rmgr::ssim::naive::retrieve_tile(b - (kRadius * ts + kRadius), tw, th, ts, kRadius, tileX, tileY, jb.b, jb.width, jb.height, jb.bStep, jb.bStride);

    Float* a2 = buf[3] + off;
    Float* b2 = buf[4] + off;
    Float* ab = buf[5] + off;
    rmgr::ssim::avx::g_multiplyFct(a2, a, a, tw, th, ts, kRadius);
    rmgr::ssim::avx::g_multiplyFct(b2, b, b, tw, th, ts, kRadius);
    rmgr::ssim::avx::g_multiplyFct(ab, a, b, tw, th, ts, kRadius);

    Float* muA = buf[0] + off;
    Float* muB = buf[1] + off;
    Float* sA2 = buf[2] + off;
    Float* sB2 = buf[3] + off;
    Float* sAB = buf[4] + off;
    jb.blur(muA, ts, a,  ts, tw, th, NULL, kRadius);
    jb.blur(muB, ts, b,  ts, tw, th, NULL, kRadius);
    jb.blur(sA2, ts, a2, ts, tw, th, NULL, kRadius);
    jb.blur(sB2, ts, b2, ts, tw, th, NULL, kRadius);
    jb.blur(sAB, ts, ab, ts, tw, th, NULL, kRadius);

    const Float c1 = Float((0.01 * 255.0) * (0.01 * 255.0));
    const Float c2 = Float((0.03 * 255.0) * (0.03 * 255.0));
    float* mapTile = jb.map ? jb.map + tileX * jb.mapStep + tileY * jb.mapStride : NULL;
    return rmgr::ssim::avx::g_sumTileFct(tw, th, ts, c1, c2, muA, muB, sA2, sB2, sAB, mapTile, jb.mapStep, jb.mapStride);
}

} // namespace

extern "C" {

/* impl: 5 = FMA blur, 4 = AVX blur (values of rmgr::ssim::Implementation, src/ssim_internal.h:41-51).
 * The AVX sum_tile supports only mapStep==1 (src/ssim.cpp:952-953); callers keep to that.
 * threads>1: OpenMP over tiles with per-thread scratch and per-thread fp64 partials, as
 * src/ssim.cpp:902-926 + src/ssim-openmp.c:26-37 do. */
int ref_compute_ssim(float* ssim, double* sumOut, uint32_t width, uint32_t height,
                     const uint8_t* a, ptrdiff_t aStep, ptrdiff_t aStride,
                     const uint8_t* b, ptrdiff_t bStep, ptrdiff_t bStride,
                     float* map, ptrdiff_t mapStep, ptrdiff_t mapStride, int impl, int threads)
{
    if (!a || !b)
        return 22;
    if (map && mapStep != 1)
        return 22;
    Job jb = {width, height, a, aStep, aStride, b, bStep, bStride, map, map ? mapStep : 0, map ? mapStride : 0,
              impl == 4 ? rmgr::ssim::avx::g_gaussianBlurFct : rmgr::ssim::fma::g_gaussianBlurFct};
    const uint32_t tilesX = (width + kTileW - 1) / kTileW;
    const uint32_t tilesY = (height + kTileH - 1) / kTileH;
    const int64_t  tiles  = int64_t(tilesX) * tilesY;
    if (threads < 1)
        threads = 1;
    threads = std::min(threads, 64); // src/ssim.cpp:1025
    std::vector<double> partial(threads, 0.0);
    int err = 0;
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        const int tn = omp_get_thread_num();
#else
        const int tn = 0;
#endif
        void* mem = NULL;
        if (posix_memalign(&mem, 64, 6 * kBufCapacity * sizeof(Float)) != 0) {
#pragma omp atomic write
            err = 12;
        } else {
            memset(mem, 0, 6 * kBufCapacity * sizeof(Float));
            double local = 0.0;
#pragma omp for schedule(static)
            for (int64_t t = 0; t < tiles; ++t)
                local += run_tile(jb, static_cast<Float*>(mem), uint32_t(t % tilesX) * kTileW, uint32_t(t / tilesX) * kTileH);
            partial[tn] = local;
            free(mem);
        }
    }
    if (err)
        return err;
    double sum = 0.0;
    for (int t = 0; t < threads; ++t)
        sum += partial[t];
    if (sumOut)
        *sumOut = sum;
    if (ssim)
        *ssim = float(sum / double(width * height));
    return 0;
}

/* The reference's own test oracle, tests/ssim_naive.h:230-339, F = double. */
double ref_naive_f64(uint32_t width, uint32_t height,
                     const uint8_t* a, ptrdiff_t aStep, ptrdiff_t aStride,
                     const uint8_t* b, ptrdiff_t bStep, ptrdiff_t bStride,
                     double* map, ptrdiff_t mapStep, ptrdiff_t mapStride)
{
    return rmgr::ssim::naive::compute_ssim<double, uint8_t>(width, height, a, aStep, aStride, b, bStep, bStride, map, mapStep, mapStride);
}

int ref_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

} // extern "C"
