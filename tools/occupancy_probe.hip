// occupancy_probe.hip -- what a gfx950 SIMD sustains in packed-fp32 instructions at a FORCED occupancy.
//
// tools/valu_probe.hip and tools/bank_probe.hip launch "CUs x 4 x W" small workgroups and assume that the dispatcher
// spreads them W per SIMD; with 30-register kernels nothing forces it to.  Here the kernel's register footprint is
// padded (a clobbered high register raises the count in the kernel descriptor) so that the HARDWARE cannot place more
// than W waves on a SIMD, and the grid is exactly the chip's capacity at that occupancy: every SIMD holds exactly W
// waves for the whole run -- the situation of ssim_strip2_kernel (230 VGPRs: W = 2; MODE_SEPARABLE 150: W = 3).
//
// Streams (per wave; 24 independent packed accumulators unless noted):
//   indep     v_pk_fma_f32 acc[j] = a * b + acc[j]            no dependency closer than 24 instructions
//   chain6x2  two interleaved dependent chains of 6 (the row sums of the blur: t = s0*k0; t = fma(s_i, k_i, t) ...)
//   chain6x1  one dependent chain of 6, back to back (what the compiler pads with s_nop)
//   ringadd   v_pk_add_f32 acc[j] = s + acc[j+1]               the ring scatter's three-address shift
//   lds       indep with one ds_read_b128 per 12 packed instructions (the window reads of the row loop)
//
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 tools/occupancy_probe.hip -o /tmp/occupancy_probe && /tmp/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

enum { S_INDEP = 0, S_CHAIN2, S_CHAIN1, S_RINGADD, S_LDS, S_FMA32, S_CVTUB, S_RCP, S_CVTF64, S_ADDF64, S_LDS1, S_LDS2, S_LDS4, S_LDS2_B64, S_VMEM2, S_VMEM4, S_LDSW4, S_LDSW4_B32, S_LDSW4_B128, S_LDSW4_2B64, S_LDSW4_2B32, S_COUNT };
static const char* kStream[] = {"indep pk_fma", "chain6 x2 interleaved", "chain6 x1 back to back", "ring pk_add shift", "indep pk_fma + ds_read_b128/12 (+ uses)",
                                "v_fma_f32 (unpacked)", "v_cvt_f32_ubyte0", "v_rcp_f32", "v_cvt_f64_f32", "v_add_f64",
                                "24 pk_fma + 1 ds_read_b128 (no use)", "24 pk_fma + 2 ds_read_b128 (no use)", "24 pk_fma + 4 ds_read_b128 (no use)", "24 pk_fma + 4 ds_read_b64 (no use)",
                                "24 pk_fma + 2 global_load_ubyte (no use)", "24 pk_fma + 4 global_load_ubyte (no use)", "24 pk_fma + 4 ds_write_b64", "24 pk_fma + 4 ds_write_b32", "24 pk_fma + 4 ds_write_b128", "24 pk_fma + 4 ds_write2_b64", "24 pk_fma + 4 ds_write2_b32"};

// W waves per SIMD -> the register the kernel pretends to use
template <int W> __device__ __forceinline__ void pad_registers()
{
    if constexpr (W == 1)      asm volatile("" ::: "v255", "a255");   // 512 unified registers: one wave per SIMD
    else if constexpr (W == 2) asm volatile("" ::: "v250");           // 256 >= n > 168
    else if constexpr (W == 3) asm volatile("" ::: "v165");           // 168 >= n > 128
    else if constexpr (W == 4) asm volatile("" ::: "v125");           // 128 >= n > 96
    else if constexpr (W == 5) asm volatile("" ::: "v95");            // 96 >= n > 80
    // W == 8: 64 or fewer, nothing to pad
}

template <int W, int STREAM>
__global__ __launch_bounds__(64) void probe(float* out, int iters, float seed)
{
    pad_registers<W>();
    __shared__ __attribute__((aligned(16))) f4 lds[64 * 4];     // 4 KiB
    constexpr int N = 24;
    f2 acc[N + 1];
#pragma unroll
    for (int j = 0; j <= N; ++j) acc[j] = f2{seed + threadIdx.x * 1e-3f + j, seed - j};
    f2 a = {1.0000001f, 0.9999999f}, b = {1e-7f, -1e-7f};
    lds[threadIdx.x] = f4{seed, seed, seed, seed};
    f4 w = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        if constexpr (STREAM == S_INDEP) {
#pragma unroll
            for (int j = 0; j < N; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
        } else if constexpr (STREAM == S_CHAIN2) {
            // 2 chains x 6, twice: 24 instructions; each instruction depends on the one two before it
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f2 t0, t1;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(acc[12 * h]), "v"(a));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(acc[12 * h + 6]), "v"(a));
#pragma unroll
                for (int k = 1; k < 6; ++k) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t0) : "v"(acc[12 * h + k]), "v"(b));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t1) : "v"(acc[12 * h + 6 + k]), "v"(b));
                }
                acc[12 * h] = t0; acc[12 * h + 6] = t1;
            }
        } else if constexpr (STREAM == S_CHAIN1) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                f2 t0;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(acc[6 * h]), "v"(a));
#pragma unroll
                for (int k = 1; k < 6; ++k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t0) : "v"(acc[6 * h + k]), "v"(b));
                acc[6 * h] = t0;
            }
        } else if constexpr (STREAM == S_RINGADD) {
#pragma unroll
            for (int j = 0; j < N; ++j) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[j]) : "v"(b), "v"(acc[j + 1]));
        } else if constexpr (STREAM == S_FMA32) {
#pragma unroll
            for (int j = 0; j < N; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].x) : "v"(a.x), "v"(b.x));
        } else if constexpr (STREAM == S_CVTUB) {
#pragma unroll
            for (int j = 0; j < N; ++j) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(acc[j].x) : "v"(acc[j].y));
        } else if constexpr (STREAM == S_RCP) {
#pragma unroll
            for (int j = 0; j < N; ++j) asm volatile("v_rcp_f32 %0, %1" : "=v"(acc[j].x) : "v"(acc[j].y));
        } else if constexpr (STREAM == S_CVTF64) {
#pragma unroll
            for (int j = 0; j < N; ++j) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(acc[j]) : "v"(a.x));
        } else if constexpr (STREAM == S_ADDF64) {
#pragma unroll
            for (int j = 0; j < N; ++j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[j]) : "v"(b));
        } else if constexpr (STREAM == S_LDS1 || STREAM == S_LDS2 || STREAM == S_LDS4 || STREAM == S_LDS2_B64) {
            // LDS reads whose results nobody uses, issued between the packed instructions; one wait per iteration.  What they cost
            // the VALU stream is what the LDS return path takes from the SIMD (register-file write port, issue slots).
            constexpr int NR = STREAM == S_LDS1 ? 1 : STREAM == S_LDS2 ? 2 : 4;
            f4 r[4]; f2 r2[4];
            const unsigned addr = (threadIdx.x * 16u) & 4095u;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (j < NR) {       // all reads at the top of the iteration: >= 20 packed instructions (x the other waves) before the wait
                    if constexpr (STREAM == S_LDS2_B64) asm volatile("ds_read_b64 %0, %1" : "=v"(r2[j]) : "v"(addr / 2));
                    else                                asm volatile("ds_read_b128 %0, %1" : "=v"(r[j]) : "v"(addr));
                }
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < NR; ++k) { if constexpr (STREAM == S_LDS2_B64) asm volatile("" :: "v"(r2[k])); else asm volatile("" :: "v"(r[k])); }
        } else if constexpr (STREAM == S_VMEM2 || STREAM == S_VMEM4) {
            // byte loads (the pixel fetches of the row loop: a wave-uniform base + a lane offset, L2 hits here) between the packed instructions
            constexpr int NR = STREAM == S_VMEM2 ? 2 : 4;
            unsigned r[4];
            const unsigned off = threadIdx.x + 64u * (i & 15);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (j < NR) asm volatile("global_load_ubyte %0, %1, %2" : "=v"(r[j]) : "v"(off), "s"(out));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < NR; ++k) asm volatile("" :: "v"(r[k]));
        } else if constexpr (STREAM == S_LDSW4 || STREAM == S_LDSW4_B32 || STREAM == S_LDSW4_B128 || STREAM == S_LDSW4_2B64 || STREAM == S_LDSW4_2B32) {
            const unsigned addr = (threadIdx.x * (STREAM == S_LDSW4_B128 ? 16u : STREAM == S_LDSW4_B32 || STREAM == S_LDSW4_2B32 ? 4u : 8u)) & 1023u;
            const f4 a4 = {a.x, a.y, b.x, b.y};
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (j < 4) {
                    if constexpr (STREAM == S_LDSW4)           asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(a) : "memory");
                    else if constexpr (STREAM == S_LDSW4_B32)  asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(a.x) : "memory");
                    else if constexpr (STREAM == S_LDSW4_B128) asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(a4) : "memory");
                    else if constexpr (STREAM == S_LDSW4_2B64) asm volatile("ds_write2_b64 %0, %1, %2 offset1:144" :: "v"(addr), "v"(a), "v"(b) : "memory");
                    else                                       asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" :: "v"(addr), "v"(a.x), "v"(a.y) : "memory");
                }
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (j % 12 == 0) { w += lds[(threadIdx.x + j) & 255]; }
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
            }
        }
    }
    f2 s = {w.x + w.z, w.y + w.w};
#pragma unroll
    for (int j = 0; j <= N; ++j) s += acc[j];
    if (s.x == 12345.678f) out[threadIdx.x] = s.x + s.y;
}

template <int W, int STREAM>
int run(float* d_out, int cus, double ghz)
{
    const int iters = 20000, blocks = cus * 4 * W;
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe<W, STREAM>, 64, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((probe<W, STREAM>), dim3(blocks), dim3(64), 0, 0, d_out, iters, 1.0f);     // clock settle
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int k = 0; k < 3; ++k) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<W, STREAM>), dim3(blocks), dim3(64), 0, 0, d_out, iters, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double clk = best * 1e-3 * ghz * 1e9 / (24.0 * iters * W);
    printf("%-32s forced %d waves/SIMD (runtime says %2d blocks/CU): %8.3f ms  %.2f clk per packed instruction per SIMD at %.3f GHz = %5.1f %% of the 4.0 clk issue peak, %5.1f T lane-ops/s\n",
           kStream[STREAM], W, occ, best, clk, ghz, 400.0 / clk, 24.0 * iters * 128.0 * blocks / (best * 1e-3) / 1e12);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return 0;
}

template <int STREAM> int sweep(float* d_out, int cus, double ghz)
{
    if (run<1, STREAM>(d_out, cus, ghz)) return 1;
    if (run<2, STREAM>(d_out, cus, ghz)) return 1;
    if (run<3, STREAM>(d_out, cus, ghz)) return 1;
    if (run<4, STREAM>(d_out, cus, ghz)) return 1;
    if (run<5, STREAM>(d_out, cus, ghz)) return 1;
    if (run<8, STREAM>(d_out, cus, ghz)) return 1;
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const double ghz = 2.375;     // the sustained clock under this load (profiles/r01_final_exact_4k_pmc_steady.md)
    printf("# %s %s, %d CUs, nominal %d kHz; clk figures assume %.3f GHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, ghz);
    float* d_out;
    CHECK(hipMalloc(&d_out, 4096));
    const int cus = p.multiProcessorCount;
    if (sweep<S_INDEP>(d_out, cus, ghz)) return 1;
    if (sweep<S_CHAIN2>(d_out, cus, ghz)) return 1;
    if (sweep<S_CHAIN1>(d_out, cus, ghz)) return 1;
    if (sweep<S_RINGADD>(d_out, cus, ghz)) return 1;
    if (sweep<S_LDS>(d_out, cus, ghz)) return 1;
    if (sweep<S_FMA32>(d_out, cus, ghz)) return 1;
    if (sweep<S_CVTUB>(d_out, cus, ghz)) return 1;
    if (sweep<S_RCP>(d_out, cus, ghz)) return 1;
    if (sweep<S_CVTF64>(d_out, cus, ghz)) return 1;
    if (sweep<S_ADDF64>(d_out, cus, ghz)) return 1;
    if (sweep<S_LDS1>(d_out, cus, ghz)) return 1;
    if (sweep<S_LDS2>(d_out, cus, ghz)) return 1;
    if (sweep<S_LDS4>(d_out, cus, ghz)) return 1;
    if (sweep<S_LDS2_B64>(d_out, cus, ghz)) return 1;
    if (sweep<S_VMEM2>(d_out, cus, ghz)) return 1;
    if (sweep<S_VMEM4>(d_out, cus, ghz)) return 1;
    if (sweep<S_LDSW4>(d_out, cus, ghz)) return 1;
    if (sweep<S_LDSW4_B32>(d_out, cus, ghz)) return 1;
    if (sweep<S_LDSW4_B128>(d_out, cus, ghz)) return 1;
    if (sweep<S_LDSW4_2B64>(d_out, cus, ghz)) return 1;
    if (sweep<S_LDSW4_2B32>(d_out, cus, ghz)) return 1;
    return 0;
}
