// A C++17 program that uses the parts of the standard library the product's own objects also instantiate (shared_ptr
// control blocks, std::thread, unique_lock, vector, exceptions under noexcept) AND the rmgr::ssim API: linked against the
// static archive librmgr-ssim.a it shares COMDAT group signatures with the archive's one relocatable object
// (a plain `ld -r` + symbol localisation left the archive unlinkable for exactly such programs).
// Prints "<errno of the call> <sum>"; without a device the errno is ENODEV.
#include <rmgr/ssim.h>
#include <rmgr/ssim-openmp.h>

#include <chrono>
#include <cstdio>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

struct Frame {
    std::vector<unsigned char> pixels;
    unsigned width, height;
    Frame(unsigned w, unsigned h, unsigned char v) : pixels(size_t(w) * h, v), width(w), height(h) {}
};

static void nap() noexcept { std::this_thread::sleep_for(std::chrono::milliseconds(1)); }

static int checked(int v)
{
    if (v < 0) throw std::runtime_error("negative: " + std::to_string(v));
    return v;
}

int main()
{
    std::shared_ptr<Frame> a = std::make_shared<Frame>(64u, 48u, (unsigned char)7);
    std::shared_ptr<Frame> b = std::make_shared<Frame>(64u, 48u, (unsigned char)9);
    std::mutex m;
    std::unique_lock<std::mutex> first(m), second;
    second = std::move(first);
    std::thread t(nap);
    t.join();

    rmgr::ssim::GeneralParams params;
    params.width = a->width;
    params.height = a->height;
    params.imgA.init_interleaved(a->pixels.data(), a->width, 1, 0);
    params.imgB.init_interleaved(b->pixels.data(), b->width, 1, 0);
    params.ssimMap = NULL;
    params.ssimStep = params.ssimStride = 0;
    params.alloc = NULL;
    params.dealloc = NULL;
    float ssim = 0.0f;
    int rc = 0, sum = 0;
    try {
        rc = checked(rmgr::ssim::compute_ssim(&ssim, params));
        sum = checked(int(a->pixels[5]) + int(b->pixels[5]));
    } catch (const std::exception& e) {
        std::printf("exception %s\n", e.what());
        return 2;
    }
    std::printf("%d %d\n", rc, sum);
    return 0;
}
