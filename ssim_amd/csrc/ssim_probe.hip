// ssim_probe.hip -- profiling aid behind rmgr_ssim_hip_probe_valu(): what the vector ALUs of THIS device sustain, right now, at a FORCED
// occupancy.  Not on the product path: no entry point that computes an SSIM launches it.
//
// Why it is in the library.  The strip kernels are fp32-VALU bound: their yardstick is the rate at which a SIMD retires
// packed fp32 instructions at the occupancy the kernel's registers allow -- two waves per SIMD for the bit-exact kernels (110 accumulator
// VGPRs), three for MODE_SEPARABLE / MODE_DOUBLE.  Rounds 4-5 took that rate from one run of tools/occupancy_probe.hip on one box (65.1 / 74.6 T
// lane-ops/s at 2 / 8 waves) and divided every later kernel time, measured on other boxes, by it; the boxes of the pool differ by +-4 %
// (profiles/r05_box_spread.md), more than any kernel change of rounds 4-5 was worth.  bench.py now runs this probe in-process before the
// clock-settle loop and right after the timed steps and reports the kernel against the SAME box's peak in the SAME run -- together with the shader
// clock both really ran at (workgroup 0 reads s_memtime / s_memrealtime), because boxes also differ in the clock they hold under the SSIM kernel's load.
//
// How the occupancy is forced: the kernel's register footprint is padded (a clobbered high register raises the VGPR count in the kernel
// descriptor) so that the hardware cannot place more than W waves on a SIMD, and the grid is exactly the chip's capacity at that occupancy
// (CUs x 4 SIMDs x W single-wave workgroups): every SIMD holds W waves from the first instruction to the last.
//
// Streams (per wave, 24 packed accumulators):
//   0  independent   v_pk_fma_f32 acc[j] = a * b + acc[j]: no dependency closer than 24 instructions -- the issue peak at that occupancy
//   1  row-sum shape two interleaved dependent chains of six (t = s0 * k0; t = fma(s_i, k_i, t) ...), the blur's row sums
#include "ssim_kernels.h"
#include <algorithm>

namespace ssim_hip {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

// W waves per SIMD -> the highest register the kernel pretends to use (512 unified registers per SIMD lane, allocated in blocks of 8).  The footprints leave
// slack: W of them fit with room to spare, W + 1 do not.  An EXACT fit (two waves of 256: what tools/occupancy_probe.hip used) was bimodal in bench.py's process --
// 67.5 T or 52 T (the ONE-wave rate) at the same measured clock, persistently within a call, depending on what had run before: the second 256-register block does not
// always find a contiguous home (profiles/r06_probe_bimodal.txt).
template <int W> __device__ __forceinline__ void pad_registers()
{
    if constexpr (W == 1)      asm volatile("" ::: "v255", "a255");   // 512: one wave per SIMD
    else if constexpr (W == 2) asm volatile("" ::: "v227");           // 2 x 232 = 464 (the bit-exact strip kernel's own footprint: 224...230); a third needs 696
    else if constexpr (W == 3) asm volatile("" ::: "v150");           // 3 x 152 = 456 (MODE_SEPARABLE: 146); a fourth needs 608
    else if constexpr (W == 4) asm volatile("" ::: "v110");           // 4 x 112 = 448; a fifth needs 560
    // W == 8: 64 or fewer, nothing to pad
}

enum { PROBE_ACCS = 24 };

template <int W, int STREAM>
__global__ __launch_bounds__(64) void probe_valu_kernel(float* out, int iters, float seed, uint64_t* clock, unsigned xcds)
{
    pad_registers<W>();
    // the shader clock this launch runs at on every XCD, as the strip kernels report theirs (ssim_kernels.hip clock_begin / clock_end): the first `xcds` workgroups
    // -- one per XCD -- add their cycles and reference ticks to per-XCD counters
    uint64_t cycles0 = 0, ticks0 = 0;
    const bool clocked = clock != nullptr && blockIdx.x < xcds;
    if (clocked) { cycles0 = __builtin_readcyclecounter(); ticks0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int N = PROBE_ACCS;
    f2 acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = f2{seed + threadIdx.x * 1e-3f + j, seed - j};
    const f2 a = {1.0000001f, 0.9999999f}, b = {1e-7f, -1e-7f};
    for (int i = 0; i < iters; ++i) {
        if constexpr (STREAM == 0) {
#pragma unroll
            for (int j = 0; j < N; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
        } else {
            // 2 chains x 6, twice: 24 instructions, each depending on the one two before it
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
        }
    }
    f2 s = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < N; ++j) s += acc[j];
    if (s.x == 12345.678f) out[threadIdx.x] = s.x + s.y;      // never true: keeps the accumulators alive
    if (clocked && threadIdx.x == 0) {
        uint64_t* c = clock + kClockStride * blockIdx.x;
        atomicAdd(reinterpret_cast<unsigned long long*>(c + 2), (unsigned long long)(__builtin_readcyclecounter() - cycles0));
        atomicAdd(reinterpret_cast<unsigned long long*>(c + 3), (unsigned long long)(__builtin_amdgcn_s_memrealtime() - ticks0));
        atomicAdd(reinterpret_cast<unsigned long long*>(c + 4), 1ull);
    }
}

template <int W>
hipError_t launch_w(int stream_kind, int blocks, float* out, int iters, hipStream_t stream, uint64_t* clock, unsigned xcds)
{
    if (stream_kind == 0) hipLaunchKernelGGL((probe_valu_kernel<W, 0>), dim3(blocks), dim3(64), 0, stream, out, iters, 1.0f, clock, xcds);
    else                  hipLaunchKernelGGL((probe_valu_kernel<W, 1>), dim3(blocks), dim3(64), 0, stream, out, iters, 1.0f, clock, xcds);
    return hipGetLastError();
}

} // namespace

uint64_t probe_valu_lane_ops(int waves_per_simd, int cu_count, int iters)
{
    // 128 lane-operations per packed instruction and wave (64 lanes x 2), PROBE_ACCS instructions per iteration
    return (uint64_t)128 * PROBE_ACCS * (uint64_t)iters * (uint64_t)cu_count * 4u * (uint64_t)waves_per_simd;
}

hipError_t launch_probe_valu(int waves_per_simd, int stream_kind, int cu_count, int xcd_count, int iters, float* out, hipStream_t stream, uint64_t* clock)
{
    const unsigned xcds = (unsigned)std::min(std::max(xcd_count, 1), (int)kClockMaxXcds);
    const int blocks = cu_count * 4 * waves_per_simd;
    switch (waves_per_simd) {
    case 1: return launch_w<1>(stream_kind, blocks, out, iters, stream, clock, xcds);
    case 2: return launch_w<2>(stream_kind, blocks, out, iters, stream, clock, xcds);
    case 3: return launch_w<3>(stream_kind, blocks, out, iters, stream, clock, xcds);
    case 4: return launch_w<4>(stream_kind, blocks, out, iters, stream, clock, xcds);
    case 8: return launch_w<8>(stream_kind, blocks, out, iters, stream, clock, xcds);
    default: return hipErrorInvalidValue;
    }
}

} // namespace ssim_hip
