// D2H probe: what bounds the host-pointer call WITH a map?  The map (4 B per pixel) travels device -> host while the next
// band's pixels (2 B per pixel) travel host -> device.  Measures: D2H into pageable and into pinned memory, H2D alone, and
// both directions at once on two streams (is the link full duplex for these copies?), for the sizes of a 4096^2 call
// (32 MiB in, 64 MiB out) in one piece and in 8 MiB chunks.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/d2h_probe tools/d2h_probe.hip -lpthread ; run on the GPU box
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main()
{
    const size_t in_bytes = size_t(32) << 20, out_bytes = size_t(64) << 20;
    char* in_pageable = (char*)malloc(in_bytes);  memset(in_pageable, 3, in_bytes);
    char* out_pageable = (char*)malloc(out_bytes); memset(out_pageable, 0, out_bytes);
    char *in_pinned, *out_pinned, *dev_in, *dev_out;
    CK(hipHostMalloc(&in_pinned, in_bytes, hipHostMallocDefault));   memset(in_pinned, 1, in_bytes);
    CK(hipHostMalloc(&out_pinned, out_bytes, hipHostMallocDefault)); memset(out_pinned, 0, out_bytes);
    CK(hipMalloc(&dev_in, in_bytes)); CK(hipMalloc(&dev_out, out_bytes));
    CK(hipMemset(dev_out, 7, out_bytes));
    hipStream_t s_in, s_out;
    CK(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
    auto best = [&](auto fn) { double b = 1e9; for (int i = 0; i < 8; ++i) { double t0 = now(); fn(); double t = now() - t0; if (t < b) b = t; } return b; };
    auto report = [&](const char* what, double t, size_t bytes) { printf("%-78s: %7.3f ms  %6.1f GB/s\n", what, t * 1e3, bytes / t / 1e9); };
    double t;
    t = best([&] { CK(hipMemcpyAsync(out_pageable, dev_out, out_bytes, hipMemcpyDeviceToHost, s_out)); CK(hipStreamSynchronize(s_out)); });
    report("D2H 64 MiB -> pageable, one hipMemcpyAsync", t, out_bytes);
    t = best([&] { CK(hipMemcpyAsync(out_pinned, dev_out, out_bytes, hipMemcpyDeviceToHost, s_out)); CK(hipStreamSynchronize(s_out)); });
    report("D2H 64 MiB -> pinned, one hipMemcpyAsync", t, out_bytes);
    t = best([&] { for (size_t o = 0; o < out_bytes; o += size_t(8) << 20) CK(hipMemcpyAsync(out_pageable + o, dev_out + o, size_t(8) << 20, hipMemcpyDeviceToHost, s_out)); CK(hipStreamSynchronize(s_out)); });
    report("D2H 64 MiB -> pageable, 8 x 8 MiB", t, out_bytes);
    t = best([&] { CK(hipMemcpyAsync(dev_in, in_pageable, in_bytes, hipMemcpyHostToDevice, s_in)); CK(hipStreamSynchronize(s_in)); });
    report("H2D 32 MiB <- pageable, one hipMemcpyAsync", t, in_bytes);
    // both directions at once
    t = best([&] {
        CK(hipMemcpyAsync(dev_in, in_pageable, in_bytes, hipMemcpyHostToDevice, s_in));
        CK(hipMemcpyAsync(out_pageable, dev_out, out_bytes, hipMemcpyDeviceToHost, s_out));
        CK(hipStreamSynchronize(s_in)); CK(hipStreamSynchronize(s_out)); });
    report("H2D 32 MiB <- pageable  +  D2H 64 MiB -> pageable at once (bytes = both)", t, in_bytes + out_bytes);
    t = best([&] {
        CK(hipMemcpyAsync(dev_in, in_pinned, in_bytes, hipMemcpyHostToDevice, s_in));
        CK(hipMemcpyAsync(out_pinned, dev_out, out_bytes, hipMemcpyDeviceToHost, s_out));
        CK(hipStreamSynchronize(s_in)); CK(hipStreamSynchronize(s_out)); });
    report("H2D 32 MiB <- pinned  +  D2H 64 MiB -> pinned at once (bytes = both)", t, in_bytes + out_bytes);
    t = best([&] {
        for (int k = 0; k < 8; ++k) {
            CK(hipMemcpyAsync(dev_in + k * (size_t(4) << 20), in_pageable + k * (size_t(4) << 20), size_t(4) << 20, hipMemcpyHostToDevice, s_in));
            CK(hipMemcpyAsync(out_pageable + k * (size_t(8) << 20), dev_out + k * (size_t(8) << 20), size_t(8) << 20, hipMemcpyDeviceToHost, s_out));
        }
        CK(hipStreamSynchronize(s_in)); CK(hipStreamSynchronize(s_out)); });
    report("the same in 8 chunks per direction, issued alternately from ONE thread (pageable)", t, in_bytes + out_bytes);
    return 0;
}
