// slot_probe.hip -- WHERE do the two waves of a SIMD sit, and does it matter?  (round 6: rmgr_ssim_hip_probe_valu at two waves per SIMD read 67.5 T or 52 T --
// the ONE-wave rate -- at the same measured clock, persistently within a call, depending on what had run before; four waves 71 or 62; three and eight were stable.)
// Every wavefront of a forced-occupancy v_pk_fma_f32 launch records the HW_ID register (wave buffer slot, SIMD, CU, SE), XCC_ID and its own start / end s_memtime; the
// host groups the waves by SIMD and prints, per launch: the kernel time, how many SIMDs hold 0 / 1 / 2 / 3+ waves at once (overlapping in time), and the mean
// per-wave duration by (slot of wave A, slot of wave B).  Between launches a perturbation kernel of a varying number of small workgroups runs, to move whatever
// state decides the placement.
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 tools/slot_probe.hip -o /tmp/slot_probe && /tmp/slot_probe [waves=2] [launches=12]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

struct Rec { unsigned hw_id, xcc; unsigned long long t0, t1; };

template <int W> __device__ __forceinline__ void pad_registers()
{
    if constexpr (W == 2)      asm volatile("" ::: "v227");
    else if constexpr (W == 3) asm volatile("" ::: "v150");
    else if constexpr (W == 4) asm volatile("" ::: "v110");
}

template <int W>
__global__ __launch_bounds__(64) void probe(Rec* rec, float* out, int iters, float seed)
{
    pad_registers<W>();
    const unsigned long long t0 = __builtin_readcyclecounter();
    constexpr int N = 24;
    f2 acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = f2{seed + threadIdx.x * 1e-3f + j, seed - j};
    const f2 a = {1.0000001f, 0.9999999f}, b = {1e-7f, -1e-7f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
    }
    f2 s = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < N; ++j) s += acc[j];
    if (s.x == 12345.678f) out[threadIdx.x] = s.x + s.y;
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        Rec r;
        r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, all 32 bits
        r.xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);        // XCC_ID[3:0]
        r.t0 = t0; r.t1 = t1;
        rec[blockIdx.x] = r;
    }
}

__global__ void perturb(float* out, int n) { if (threadIdx.x == 9999) out[0] = (float)n; }

template <int W> int run(int launches)
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, blocks = cus * 4 * W, iters = 40000 / W;
    Rec* d_rec; float* d_out;
    CHECK(hipMalloc(&d_rec, sizeof(Rec) * blocks));
    CHECK(hipMalloc(&d_out, 4096));
    std::vector<Rec> rec(blocks);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((probe<W>), dim3(blocks), dim3(64), 0, 0, d_rec, d_out, iters, 1.0f);      // clock settle
    CHECK(hipDeviceSynchronize());
    for (int l = 0; l < launches; ++l) {
        // perturbation: a varying number of tiny workgroups, sometimes an idle gap
        const int n = (l * 37) % 11;
        for (int k = 0; k < n; ++k) hipLaunchKernelGGL(perturb, dim3(1 + 13 * k), dim3(64), 0, 0, d_out, k);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL((probe<W>), dim3(blocks), dim3(64), 0, 0, d_rec, d_out, iters, 1.0f);      // one untimed (placement as the timed one's predecessor)
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<W>), dim3(blocks), dim3(64), 0, 0, d_rec, d_out, iters, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(rec.data(), d_rec, sizeof(Rec) * blocks, hipMemcpyDeviceToHost));
        std::map<unsigned, std::vector<int> > by_simd;          // key: xcc, se, sh, cu, simd
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int i = 0; i < blocks; ++i) {
            const unsigned h = rec[i].hw_id;
            const unsigned key = (rec[i].xcc << 16) | (((h >> 13) & 3) << 12) | (((h >> 12) & 1) << 11) | (((h >> 8) & 15) << 4) | ((h >> 4) & 3);
            by_simd[key].push_back(i);
            tmin = std::min(tmin, rec[i].t0); tmax = std::max(tmax, rec[i].t1);
        }
        std::map<int, int> per_simd_count;                      // waves per SIMD
        std::map<unsigned, std::pair<double, int> > by_slots;   // (sorted slot list) -> (sum of durations, count)
        std::map<unsigned, int> late;                           // (slot list) -> waves that started after another wave of the SIMD ended (ran in a second round)
        for (auto& kv : by_simd) {
            per_simd_count[(int)kv.second.size()]++;
            std::vector<unsigned> slots;
            for (int i : kv.second) slots.push_back(rec[i].hw_id & 15);
            std::sort(slots.begin(), slots.end());
            unsigned sk = 0;
            for (unsigned s : slots) sk = sk * 16 + s + 1;
            for (int i : kv.second) {
                by_slots[sk].first += (double)(rec[i].t1 - rec[i].t0); by_slots[sk].second++;
                for (int j : kv.second) if (j != i && rec[i].t0 >= rec[j].t1) { late[sk]++; break; }
            }
        }
        // per XCD (s_memtime is local to an XCD): how long the dispatcher took to start the XCD's waves, and how long its waves overlapped
        double start_spread = 0, run_all = 0; int nx = 0;
        for (unsigned x = 0; x < 16; ++x) {
            unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0; int cnt = 0;
            for (int i = 0; i < blocks; ++i) if (rec[i].xcc == x) { a0 = std::min(a0, rec[i].t0); a1 = std::max(a1, rec[i].t0); b0 = std::min(b0, rec[i].t1); b1 = std::max(b1, rec[i].t1); ++cnt; }
            if (!cnt) continue;
            start_spread += (double)(a1 - a0); run_all += (double)(b1 - a0); ++nx;
        }
        printf("   per XCD: last wave starts %.0f kcycles after the first; first start to last end %.0f kcycles\n", start_spread / nx / 1e3, run_all / nx / 1e3);
        printf("launch %2d (after %2d perturbation kernels): %.3f ms = %5.1f T lane-ops/s; span %.0f kcycles; SIMDs in use %zu of %d; waves per SIMD:", l, n, ms,
               24.0 * iters * 128.0 * blocks / (ms * 1e-3) / 1e12, (tmax - tmin) / 1e3, by_simd.size(), cus * 4);
        for (auto& c : per_simd_count) printf(" %dx%d", c.second, c.first);
        printf("\n");
        for (auto& c : by_slots) {
            printf("     slots {");
            std::vector<unsigned> s; unsigned v = c.first; while (v) { s.push_back(v % 16 - 1); v /= 16; }
            for (size_t i = s.size(); i-- > 0;) printf("%u%s", s[i], i ? "," : "");
            printf("}: %d waves, mean %.0f kcycles each, %d started after a SIMD mate had ended\n", c.second.second, c.second.first / c.second.second / 1e3, late[c.first]);
        }
    }
    return 0;
}

int main(int argc, char** argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 2, launches = argc > 2 ? atoi(argv[2]) : 12;
    if (waves == 2) return run<2>(launches);
    if (waves == 3) return run<3>(launches);
    if (waves == 4) return run<4>(launches);
    return run<8>(launches);
}
