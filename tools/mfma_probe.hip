// mfma_probe.hip -- can the matrix pipe serve as a second fp32 FMA port for the blur's row sums?
//
// v_mfma_f32_4x4x1_16b_f32 is 16 independent 4x4 outer products with K = 1: no contraction at all, every
// output element is ONE fused multiply-add  D[i] = A(i) * B + C[i].  With A = four taps K(t, 0..3) (one per
// lane of a quad) and B = the lane's own folded pixel sum, one instruction gives a lane four of its six row
// sums' next chain step.  This probe checks (1) the operand layout and that the result is bit-identical to
// fmaf(), (2) what the pipe sustains alone, and (3) what it costs next to scalar / packed fp32 VALU work from
// the same wave at the SSIM kernel's occupancy.
// build+run (GPU box): hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__global__ void layout_kernel(const float* a, const float* b, const f4* c, f4* d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c[l], 0, 0, 0);
}

// MODE 0: MFMA only; 1: scalar v_fma_f32 only; 2: v_pk_fma_f32 only; 3: MFMA + scalar; 4: MFMA + packed
// per iteration: NM MFMAs (independent accumulators) and NV VALU lane-op pairs
template <int MODE, int NM, int NV>
__global__ __launch_bounds__(64) void mix_kernel(float* out, int iters, float seed)
{
    f4 macc[NM > 0 ? NM : 1];
    f2 vacc[NV > 0 ? NV : 1];
#pragma unroll
    for (int j = 0; j < (NM > 0 ? NM : 1); ++j) macc[j] = f4{seed, seed + j, seed - j, seed * j};
#pragma unroll
    for (int j = 0; j < (NV > 0 ? NV : 1); ++j) vacc[j] = f2{seed + threadIdx.x * 1e-3f + j, seed - j};
    float ka = 1e-7f * (threadIdx.x & 3), kb = 1.0000001f + threadIdx.x * 1e-9f;
    f2 a = {1.0000001f, 0.9999999f}, b = {1e-7f, -1e-7f};
    for (int i = 0; i < iters; ++i) {
        constexpr int STEPS = (NM > NV ? NM : NV);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            if constexpr (MODE == 0 || MODE >= 3) {
                if (s * NM / STEPS != (s + 1) * NM / STEPS) {
                    const int j = s * NM / STEPS;
                    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(macc[j]) : "v"(ka), "v"(kb));
                }
            }
            if constexpr (MODE == 1 || MODE == 3) {
                if (s * NV / STEPS != (s + 1) * NV / STEPS) {
                    const int j = s * NV / STEPS;
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(vacc[j].x) : "v"(a.x), "v"(b.x));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(vacc[j].y) : "v"(a.y), "v"(b.y));
                }
            }
            if constexpr (MODE == 2 || MODE == 4) {
                if (s * NV / STEPS != (s + 1) * NV / STEPS) {
                    const int j = s * NV / STEPS;
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(vacc[j]) : "v"(a), "v"(b));
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < (NM > 0 ? NM : 1); ++j) s += macc[j].x + macc[j].y + macc[j].z + macc[j].w;
#pragma unroll
    for (int j = 0; j < (NV > 0 ? NV : 1); ++j) s += vacc[j].x + vacc[j].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE, int NM, int NV>
int run(const char* name, float* d_out, int cus)
{
    const int iters = 4000;
    for (int waves = 1; waves <= 4; waves *= 2) {
        const int blocks = cus * 4 * waves;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL((mix_kernel<MODE, NM, NV>), dim3(blocks), dim3(64), 0, 0, d_out, 100, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((mix_kernel<MODE, NM, NV>), dim3(blocks), dim3(64), 0, 0, d_out, iters, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const bool m = (MODE == 0 || MODE >= 3), v = (MODE != 0);
        const double mfma_lane_fma = m ? (double)blocks * 64 * 4.0 * NM * iters : 0;    // 4 fma per lane per MFMA
        const double valu_lane_ops = v ? (double)blocks * 64 * 2.0 * NV * iters : 0;
        const double clk_iter = ms * 1e-3 * 2.4e9 / iters / waves;                     // SIMD clocks per wave-iteration at 2.4 GHz
        printf("%-34s waves/SIMD %d: %8.3f ms  mfma %6.2f T fma/s  valu %6.2f T lane-ops/s  sum %6.2f  ~%.0f clk per wave-iter\n",
               name, waves, ms, mfma_lane_fma / ms / 1e9, valu_lane_ops / ms / 1e9, (mfma_lane_fma + valu_lane_ops) / ms / 1e9, clk_iter);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s %s CUs %d clock %d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);

    // ---- (1) layout + exactness ----
    {
        float ha[64], hb[64];
        f4 hc[64], hd[64];
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) * (1.0f / 977.0f) + 1e-3f; };
        for (int l = 0; l < 64; ++l) {
            ha[l] = rnd() * 1e-3f; hb[l] = rnd() * 37.0f;
            hc[l] = f4{rnd(), rnd() * 100.0f, rnd() * 1e-2f, rnd() * 7.0f};
        }
        float *da, *db; f4 *dc, *dd;
        CHECK(hipMalloc(&da, sizeof(ha))); CHECK(hipMalloc(&db, sizeof(hb))); CHECK(hipMalloc(&dc, sizeof(hc))); CHECK(hipMalloc(&dd, sizeof(hd)));
        CHECK(hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dc, hc, sizeof(hc), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        CHECK(hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost));
        int okP = 0, okQ = 0, okUnfused = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const float got = hd[l][i];
                const float p1 = fmaf(ha[4 * (l / 4) + i], hb[l], hc[l][i]);       // D[i] of lane (b,j) = A(lane 4b+i) * B(own)
                const float p2 = fmaf(ha[l], hb[4 * (l / 4) + i], hc[l][i]);       // transposed guess
                const float p3 = ha[4 * (l / 4) + i] * hb[l] + hc[l][i];           // unfused
                okP += memcmp(&got, &p1, 4) == 0; okQ += memcmp(&got, &p2, 4) == 0; okUnfused += memcmp(&got, &p3, 4) == 0;
            }
        printf("layout: D[i]@lane(b,j) == fmaf(A@lane(4b+i), B@own, C[i]) bitwise: %d/256; transposed model: %d/256; unfused model: %d/256\n", okP, okQ, okUnfused);
    }

    float* d_out;
    CHECK(hipMalloc(&d_out, 4096));
    const int cus = p.multiProcessorCount;
    if (run<0, 12, 0>("mfma only (12/iter)", d_out, cus)) return 1;
    if (run<1, 0, 32>("v_fma_f32 only (64/iter)", d_out, cus)) return 1;
    if (run<2, 0, 32>("v_pk_fma_f32 only (32/iter)", d_out, cus)) return 1;
    if (run<3, 12, 32>("12 mfma + 64 v_fma_f32", d_out, cus)) return 1;
    if (run<4, 12, 32>("12 mfma + 32 v_pk_fma_f32", d_out, cus)) return 1;
    if (run<3, 16, 32>("16 mfma + 64 v_fma_f32", d_out, cus)) return 1;
    if (run<3, 8, 32>("8 mfma + 64 v_fma_f32", d_out, cus)) return 1;
    if (run<4, 8, 32>("8 mfma + 32 v_pk_fma_f32", d_out, cus)) return 1;
    return 0;
}
