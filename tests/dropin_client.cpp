// A client written against the reference's public API only (rmgr/ssim.h, rmgr/ssim-openmp.h), the
// way tests/rmgr-ssim-tests.cpp:228-336 and sample/rmgr-ssim-sample.cpp:78-101 of romigrou/ssim
// use it: deprecated Params block, init_interleaved, compute_ssim(params), get_errno, map output.
// Built with -std=c++98 and linked against librmgr-ssim-hip.so by tests/test_abi_cpu.py.
//
// usage: dropin_client a.u8 b.u8 width height channels   -> prints one line per channel
#include <rmgr/ssim.h>
#include <rmgr/ssim-openmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

// A caller-supplied thread pool as the reference's tests write one (tests/rmgr-ssim-tests.cpp: a serial loop over the jobs); `context` counts the jobs,
// a negative count makes the pool report failure after running them (src/ssim.cpp:1094-1097: ECHILD).
static rmgr::ssim::int32_t serial_pool(void* context, rmgr::ssim::ThreadFct fct, void* const args[], rmgr::ssim::uint32_t threadCount, rmgr::ssim::uint32_t jobCount) RMGR_NOEXCEPT
{
    int* counter = static_cast<int*>(context);
    const bool fail = *counter < 0;
    if (threadCount == 0) return 1;
    for (rmgr::ssim::uint32_t job = 0; job < jobCount; ++job)
        fct(args[0], job);
    *counter = int(jobCount);
    return fail ? 1 : 0;
}

static bool read_file(const char* path, std::vector<unsigned char>& out, size_t n)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    out.resize(n);
    const size_t got = fread(&out[0], 1, n, f);
    fclose(f);
    return got == n;
}

int main(int argc, char** argv)
{
    if (argc < 6) {
        fprintf(stderr, "usage: %s a.u8 b.u8 width height channels\n", argv[0]);
        return 2;
    }
    const int width = atoi(argv[3]), height = atoi(argv[4]), channels = atoi(argv[5]);
    std::vector<unsigned char> a, b;
    if (!read_file(argv[1], a, size_t(width) * height * channels) || !read_file(argv[2], b, size_t(width) * height * channels)) {
        fprintf(stderr, "cannot read inputs\n");
        return 2;
    }
    const rmgr::ssim::Version ver = rmgr::ssim::get_version();
    printf("version %u.%u.%u %s\n", ver.major, ver.minor, ver.patch, ver.string);

    std::vector<float> map(size_t(width) * height);
    rmgr::ssim::Params params = rmgr::ssim::Params();   // value-initialised: all zero
    params.width      = width;
    params.height     = height;
    params.ssimMap    = &map[0];
    params.ssimStep   = 1;
    params.ssimStride = width;
    params.use_default_allocator();

    for (int c = 0; c < channels; ++c) {
        params.imgA.init_interleaved(&a[0], width * channels, channels, c);
        params.imgB.init_interleaved(&b[0], width * channels, channels, c);
        // the header's own diagnostics macros, used the way the reference's tests use them (tests/rmgr-ssim-tests.cpp:35-40)
        RMGR_WARNING_PUSH()
        RMGR_WARNING_MSVC_DISABLE(4996)
        RMGR_WARNING_GCC_DISABLE("-Wdeprecated-declarations")
        RMGR_WARNING_CLANG_DISABLE("-Wdeprecated-declarations")
        const float ssim = rmgr::ssim::compute_ssim(params);
        const int   err  = rmgr::ssim::get_errno(ssim);
        RMGR_WARNING_POP()
        if (err != 0) {
            printf("channel %d errno %d\n", c, err);
            continue;
        }
        float viaOmp = -1.0f;
        const int rc = rmgr::ssim::compute_ssim_openmp(&viaOmp, params);
        double mapSum = 0.0;
        for (size_t i = 0; i < map.size(); ++i) mapSum += map[i];
        union { float f; unsigned u; } bits, bitsOmp;
        bits.f = ssim; bitsOmp.f = viaOmp;
        // the deprecated Params block carries its thread pool (include/rmgr/ssim.h:692-697): the pool's dispatch function runs the jobs, the value is the same
        int jobs = 0, failing = -1;
        rmgr::ssim::Params pooled = params;
        pooled.threadPool = serial_pool; pooled.threadPoolContext = &jobs; pooled.threadCount = 3;
        RMGR_WARNING_PUSH()
        RMGR_WARNING_MSVC_DISABLE(4996)
        RMGR_WARNING_GCC_DISABLE("-Wdeprecated-declarations")
        RMGR_WARNING_CLANG_DISABLE("-Wdeprecated-declarations")
        union { float f; unsigned u; } bitsPool;
        bitsPool.f = rmgr::ssim::compute_ssim(pooled);
        pooled.threadPoolContext = &failing;
        const int failErr = rmgr::ssim::get_errno(rmgr::ssim::compute_ssim(pooled));
        RMGR_WARNING_POP()
        printf("channel %d ssim 0x%08x openmp_rc %d openmp 0x%08x map_mean %.9f pool 0x%08x jobs %d failing_pool_errno %d\n", c, bits.u, rc, bitsOmp.u, mapSum / double(map.size()),
               bitsPool.u, jobs, failErr);
    }
    return 0;
}
