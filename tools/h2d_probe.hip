// H2D staging probe: what bounds the host-pointer call?  Pageable hipMemcpy vs pinned DMA vs threaded memcpy into pinned
// memory, and a chunked pipeline (N worker threads memcpy chunk k+1 into a pinned ring while the DMA engine sends chunk k).
// build: hipcc --offload-arch=gfx950 -O2 -o build/tools/h2d_probe tools/h2d_probe.hip -lpthread ; run on the GPU box
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static void par_memcpy(char* dst, const char* src, size_t n, int threads)
{
    if (threads <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> t;
    size_t per = (n + threads - 1) / threads;
    per = (per + 4095) & ~size_t(4095);
    for (int i = 0; i < threads; ++i) {
        size_t a = i * per, b = a + per > n ? n : a + per;
        if (a >= n) break;
        t.emplace_back([=] { memcpy(dst + a, src + a, b - a); });
    }
    for (auto& x : t) x.join();
}

int main(int argc, char** argv)
{
    size_t bytes = (argc > 1 ? atol(argv[1]) : 32) << 20;
    char* pageable = (char*)malloc(bytes);
    memset(pageable, 3, bytes);
    char *pinned, *dev;
    CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
    CK(hipMalloc(&dev, bytes));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    memset(pinned, 1, bytes);
    auto best = [&](auto fn) { double b = 1e9; for (int i = 0; i < 8; ++i) { double t0 = now(); fn(); double t = now() - t0; if (t < b) b = t; } return b; };
    double t;
    t = best([&] { CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); });
    printf("%zu MiB pageable hipMemcpyAsync          : %.3f ms  %.1f GB/s\n", bytes >> 20, t * 1e3, bytes / t / 1e9);
    t = best([&] { CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); });
    printf("%zu MiB pinned   hipMemcpyAsync          : %.3f ms  %.1f GB/s\n", bytes >> 20, t * 1e3, bytes / t / 1e9);
    for (int th : {1, 2, 4, 8, 16}) {
        t = best([&] { par_memcpy(pinned, pageable, bytes, th); });
        printf("memcpy pageable -> pinned, %2d threads (spawned per call): %.3f ms  %.1f GB/s\n", th, t * 1e3, bytes / t / 1e9);
    }
    // persistent worker pool: chunked pipeline
    for (int th : {2, 4, 8}) for (size_t chunk : {size_t(2) << 20, size_t(4) << 20, size_t(8) << 20}) {
        const int nchunks = (int)((bytes + chunk - 1) / chunk);
        std::atomic<int> next_chunk(0), go(0), done_threads(0);
        std::vector<std::atomic<int>> ready(nchunks);
        std::atomic<bool> quit(false);
        std::vector<std::thread> pool;
        for (int i = 0; i < th; ++i) pool.emplace_back([&] {
            int seen = 0;
            for (;;) {
                while (go.load(std::memory_order_acquire) == seen) { if (quit.load()) return; }
                seen = go.load();
                for (;;) {
                    int k = next_chunk.fetch_add(1);
                    if (k >= nchunks) break;
                    size_t a = (size_t)k * chunk, b = a + chunk > bytes ? bytes : a + chunk;
                    memcpy(pinned + a, pageable + a, b - a);
                    ready[k].store(seen, std::memory_order_release);
                }
                done_threads.fetch_add(1);
            }
        });
        int gen = 0;
        t = best([&] {
            ++gen;
            next_chunk.store(0); done_threads.store(0);
            go.store(gen, std::memory_order_release);
            for (int k = 0; k < nchunks; ++k) {
                while (ready[k].load(std::memory_order_acquire) != gen) {}
                size_t a = (size_t)k * chunk, b = a + chunk > bytes ? bytes : a + chunk;
                CK(hipMemcpyAsync(dev + a, pinned + a, b - a, hipMemcpyHostToDevice, s));
            }
            CK(hipStreamSynchronize(s));
            while (done_threads.load() != th) {}
        });
        printf("pipeline: %d pool threads, %zu MiB chunks: %.3f ms  %.1f GB/s\n", th, chunk >> 20, t * 1e3, bytes / t / 1e9);
        quit.store(true);
        for (auto& x : pool) x.join();
    }
    return 0;
}
