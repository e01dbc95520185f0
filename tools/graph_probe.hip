// graph_probe.hip -- does a hipGraph shorten the blocking single-pair call?  The call is: strip kernel (~90 us) -> reduction kernel (~4 us) ->
// hipStreamSynchronize; round 4 measured 12 us of launch latency + wake-up and 3.5 us for the second launch around a 94 us kernel.
// This probe times, per iteration and wall clock: (a) two plain launches + synchronize, (b) one hipGraphLaunch of the same two kernel nodes +
// synchronize, (c) the same with the kernel node parameters re-set before every launch (what a per-call graph would need: the image pointers change).
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 tools/graph_probe.hip -o /tmp/graph_probe && /tmp/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void busy(float* out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0000001f;
    for (int i = 0; i < iters; ++i) a = a * b + 1e-7f;
    if (a == 123.456f) out[0] = a;
}
__global__ void tiny(float* out, float v) { if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = v; }

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main()
{
    float* d = nullptr;
    CHECK(hipMalloc(&d, 64));
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const dim3 grid(2048), block(64);
    int iters = 20000;
    // calibrate `busy` to ~90 us
    for (int rep = 0; rep < 6; ++rep) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0, s)); hipLaunchKernelGGL(busy, grid, block, 0, s, d, iters); CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) iters = (int)(iters * 0.090 / ms);
    }
    const int N = 300;
    std::vector<double> ta, tb, tc;
    for (int i = 0; i < N + 20; ++i) {
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(busy, grid, block, 0, s, d, iters);
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, 1.0f);
        CHECK(hipStreamSynchronize(s));
        if (i >= 20) ta.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    // the same two nodes as a graph
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(busy, grid, block, 0, s, d, iters);
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, 1.0f);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < N + 20; ++i) {
        auto t0 = std::chrono::steady_clock::now();
        CHECK(hipGraphLaunch(ge, s));
        CHECK(hipStreamSynchronize(s));
        if (i >= 20) tb.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    // ... with the kernel parameters of both nodes re-set before every launch
    size_t n = 0; CHECK(hipGraphGetNodes(g, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n); CHECK(hipGraphGetNodes(g, nodes.data(), &n));
    for (int i = 0; i < N + 20; ++i) {
        auto t0 = std::chrono::steady_clock::now();
        for (size_t k = 0; k < n; ++k) {
            hipKernelNodeParams p;
            CHECK(hipGraphKernelNodeGetParams(nodes[k], &p));
            CHECK(hipGraphExecKernelNodeSetParams(ge, nodes[k], &p));
        }
        CHECK(hipGraphLaunch(ge, s));
        CHECK(hipStreamSynchronize(s));
        if (i >= 20) tc.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, s)); hipLaunchKernelGGL(busy, grid, block, 0, s, d, iters); CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("busy kernel %.1f us (events); per blocking iteration, median / min of %d (wall): two launches + sync %.1f / %.1f us | graph launch + sync %.1f / %.1f us | graph with node parameters re-set %.1f / %.1f us\n",
           ms * 1e3, N, med(ta), *std::min_element(ta.begin(), ta.end()), med(tb), *std::min_element(tb.begin(), tb.end()), med(tc), *std::min_element(tc.begin(), tc.end()));
    return 0;
}
