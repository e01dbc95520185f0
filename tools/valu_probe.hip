// valu_probe.hip -- measures what the gfx950 VALU actually sustains for the instruction kinds the
// SSIM kernel is made of (the roofline the kernel is priced against in DESIGN.md).
// build+run (GPU box): hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

enum { K_FMA = 0, K_PK_FMA, K_PK_ADD, K_PK_MUL, K_ADD, K_PK_FMA_SGPR, K_MIX, K_DIV, K_FMA64, K_COUNT };
static const char* kNames[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_add_f32", "v_pk_fma_f32(sgpr,op_sel)",
                               "mix 30pkfma+15pkadd+6pkmul", "ieee fdiv f32", "v_fma_f64"};
static const double kLaneOps[] = {1, 2, 2, 2, 1, 2, 2, 1, 1};   // lane-ops per lane per instruction

template <int KIND>
__global__ __launch_bounds__(64) void probe(float* out, int iters, float seed)
{
    constexpr int N = 24;
    f2 acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = f2{seed + threadIdx.x * 1e-3f + j, seed - j};
    f2 a = {1.0000001f, 0.9999999f}, b = {1e-7f, -1e-7f};
    float ks = seed * 1e-7f;   // wave-uniform -> SGPR
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if constexpr (KIND == K_FMA) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].x) : "v"(a.x), "v"(b.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].y) : "v"(a.y), "v"(b.y));
            } else if constexpr (KIND == K_PK_FMA) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
            } else if constexpr (KIND == K_PK_ADD) {
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[j]) : "v"(b));
            } else if constexpr (KIND == K_PK_MUL) {
                asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc[j]) : "v"(a));
            } else if constexpr (KIND == K_ADD) {
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[j].x) : "v"(b.x));
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[j].y) : "v"(b.y));
            } else if constexpr (KIND == K_PK_FMA_SGPR) {
                f2 kk = {ks, ks};
                acc[j] = __builtin_elementwise_fma(a, kk, acc[j]);
            } else if constexpr (KIND == K_MIX) {
                // the blur's own ratio per 51 instructions: 30 fma, 15 add, 6 mul
                const int m = (i * N + j) % 17;
                if (m < 10)      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
                else if (m < 15) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[j]) : "v"(b));
                else             asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc[j]) : "v"(a));
            } else if constexpr (KIND == K_DIV) {
                acc[j].x = acc[j].x / (a.x + acc[j].y);
            } else if constexpr (KIND == K_FMA64) {
                double d = __builtin_bit_cast(double, acc[j]);
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d) : "v"((double)a.x), "v"((double)b.x));
                acc[j] = __builtin_bit_cast(f2, d);
            }
        }
    }
    f2 s = {0, 0};
#pragma unroll
    for (int j = 0; j < N; ++j) s += acc[j];
    if (s.x == 12345.678f) out[threadIdx.x] = s.x + s.y;   // keep everything live
}

template <int KIND>
int run(float* d_out, int cus)
{
    constexpr int N = 24;
    const double per_iter = (KIND == K_FMA || KIND == K_ADD) ? 2.0 * N : 1.0 * N;   // instructions per lane per iteration
    const int iters = (KIND == K_DIV) ? 2000 : 20000;
    for (int waves = 1; waves <= 8; waves *= 2) {
        const int blocks = cus * 4 * waves;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, 100, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, iters, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr = (double)blocks * 64 * per_iter * iters;      // lane-instructions
        const double ops = instr * ((KIND == K_FMA || KIND == K_ADD) ? 1.0 : kLaneOps[KIND]);
        const double cyc_per_wave_instr = (ms * 1e-3 * 2.4e9) / (per_iter * iters * waves);   // at 2.4 GHz nominal, per SIMD
        printf("%-28s waves/SIMD %d: %8.3f ms  %7.2f T lane-instr/s  %7.2f T lane-ops/s  ~%.2f clk per wave-instr per SIMD (2.4 GHz)\n",
               kNames[KIND], waves, ms, instr / ms / 1e9, ops / ms / 1e9, cyc_per_wave_instr);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s %s CUs %d clock %d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
    float* d_out;
    CHECK(hipMalloc(&d_out, 4096));
    const int cus = p.multiProcessorCount;
    if (run<K_FMA>(d_out, cus)) return 1;
    if (run<K_PK_FMA>(d_out, cus)) return 1;
    if (run<K_PK_ADD>(d_out, cus)) return 1;
    if (run<K_PK_MUL>(d_out, cus)) return 1;
    if (run<K_ADD>(d_out, cus)) return 1;
    if (run<K_PK_FMA_SGPR>(d_out, cus)) return 1;
    if (run<K_MIX>(d_out, cus)) return 1;
    if (run<K_DIV>(d_out, cus)) return 1;
    if (run<K_FMA64>(d_out, cus)) return 1;
    return 0;
}
