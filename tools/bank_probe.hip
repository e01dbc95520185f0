// bank_probe.hip -- does the VGPR bank (register index mod 4) of a packed instruction's operands change its
// issue cost on gfx950?  Hand-placed registers, 24 independent accumulators, 2 / 4 / 8 waves per SIMD.
// build+run (GPU box): hipcc --offload-arch=gfx950 -O3 tools/bank_probe.hip -o /tmp/bank_probe && /tmp/bank_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// Accumulators: v[16+2j : 17+2j], j = 0..23 -> bank groups alternate {0,1}, {2,3}.
// Constant operands: v[4:5] (banks {0,1}), v[6:7] (banks {2,3}); s[4:5] holds a uniform constant.
// MODE 0: pk_add, constant in the OTHER bank group than the accumulator      (no shared bank)
// MODE 1: pk_add, constant in the SAME bank group as the accumulator
// MODE 2: pk_fma acc = c * s + acc, constant vector in the OTHER group
// MODE 3: pk_fma acc = c * s + acc, constant vector in the SAME group
// MODE 4: pk_add three-address ring shift: acc[j] = c + acc[j+1]  (dst group != src group, like the kernel's scatter)
// MODE 5: pk_fma with both VGPR sources = the accumulator itself (acc = acc * s + acc)
template <int MODE>
__global__ __launch_bounds__(64) void probe(float* out, int iters)
{
    asm volatile(
        "v_mov_b32 v4, 1.0\n v_mov_b32 v5, 1.0\n v_mov_b32 v6, 1.0\n v_mov_b32 v7, 1.0\n"
        "s_mov_b32 s4, 0\n s_mov_b32 s5, 0\n" ::: "v4", "v5", "v6", "v7", "s4", "s5");
#define Z(n) "v_mov_b32 v" #n ", 0\n"
    asm volatile(Z(16) Z(17) Z(18) Z(19) Z(20) Z(21) Z(22) Z(23) Z(24) Z(25) Z(26) Z(27) Z(28) Z(29) Z(30) Z(31) Z(32) Z(33) Z(34) Z(35) Z(36) Z(37) Z(38) Z(39)
                 Z(40) Z(41) Z(42) Z(43) Z(44) Z(45) Z(46) Z(47) Z(48) Z(49) Z(50) Z(51) Z(52) Z(53) Z(54) Z(55) Z(56) Z(57) Z(58) Z(59) Z(60) Z(61) Z(62) Z(63) Z(64) Z(65)
                 ::: "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39",
                     "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65");
#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39", \
             "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65"
    for (int i = 0; i < iters; ++i) {
        // even j: accumulator in banks {0,1}; odd j: {2,3}
#define ADD(d, c)     "v_pk_add_f32 v[" #d ":" #d "+1], v[" #c ":" #c "+1], v[" #d ":" #d "+1]\n"
#define FMA(d, c)     "v_pk_fma_f32 v[" #d ":" #d "+1], v[" #c ":" #c "+1], s[4:5], v[" #d ":" #d "+1]\n"
#define SHIFT(d, s, c) "v_pk_add_f32 v[" #d ":" #d "+1], v[" #c ":" #c "+1], v[" #s ":" #s "+1]\n"
#define SELF(d)       "v_pk_fma_f32 v[" #d ":" #d "+1], v[" #d ":" #d "+1], s[4:5], v[" #d ":" #d "+1]\n"
        if constexpr (MODE == 0)
            asm volatile(ADD(16,6) ADD(18,4) ADD(20,6) ADD(22,4) ADD(24,6) ADD(26,4) ADD(28,6) ADD(30,4) ADD(32,6) ADD(34,4) ADD(36,6) ADD(38,4)
                         ADD(40,6) ADD(42,4) ADD(44,6) ADD(46,4) ADD(48,6) ADD(50,4) ADD(52,6) ADD(54,4) ADD(56,6) ADD(58,4) ADD(60,6) ADD(62,4) ::: CLOB);
        else if constexpr (MODE == 1)
            asm volatile(ADD(16,4) ADD(18,6) ADD(20,4) ADD(22,6) ADD(24,4) ADD(26,6) ADD(28,4) ADD(30,6) ADD(32,4) ADD(34,6) ADD(36,4) ADD(38,6)
                         ADD(40,4) ADD(42,6) ADD(44,4) ADD(46,6) ADD(48,4) ADD(50,6) ADD(52,4) ADD(54,6) ADD(56,4) ADD(58,6) ADD(60,4) ADD(62,6) ::: CLOB);
        else if constexpr (MODE == 2)
            asm volatile(FMA(16,6) FMA(18,4) FMA(20,6) FMA(22,4) FMA(24,6) FMA(26,4) FMA(28,6) FMA(30,4) FMA(32,6) FMA(34,4) FMA(36,6) FMA(38,4)
                         FMA(40,6) FMA(42,4) FMA(44,6) FMA(46,4) FMA(48,6) FMA(50,4) FMA(52,6) FMA(54,4) FMA(56,6) FMA(58,4) FMA(60,6) FMA(62,4) ::: CLOB);
        else if constexpr (MODE == 3)
            asm volatile(FMA(16,4) FMA(18,6) FMA(20,4) FMA(22,6) FMA(24,4) FMA(26,6) FMA(28,4) FMA(30,6) FMA(32,4) FMA(34,6) FMA(36,4) FMA(38,6)
                         FMA(40,4) FMA(42,6) FMA(44,4) FMA(46,6) FMA(48,4) FMA(50,6) FMA(52,4) FMA(54,6) FMA(56,4) FMA(58,6) FMA(60,4) FMA(62,6) ::: CLOB);
        else if constexpr (MODE == 4)
            asm volatile(SHIFT(16,18,4) SHIFT(18,20,6) SHIFT(20,22,4) SHIFT(22,24,6) SHIFT(24,26,4) SHIFT(26,28,6) SHIFT(28,30,4) SHIFT(30,32,6) SHIFT(32,34,4) SHIFT(34,36,6)
                         SHIFT(36,38,4) SHIFT(38,40,6) SHIFT(40,42,4) SHIFT(42,44,6) SHIFT(44,46,4) SHIFT(46,48,6) SHIFT(48,50,4) SHIFT(50,52,6) SHIFT(52,54,4) SHIFT(54,56,6)
                         SHIFT(56,58,4) SHIFT(58,60,6) SHIFT(60,62,4) SHIFT(62,64,6) ::: CLOB);
        else
            asm volatile(SELF(16) SELF(18) SELF(20) SELF(22) SELF(24) SELF(26) SELF(28) SELF(30) SELF(32) SELF(34) SELF(36) SELF(38)
                         SELF(40) SELF(42) SELF(44) SELF(46) SELF(48) SELF(50) SELF(52) SELF(54) SELF(56) SELF(58) SELF(60) SELF(62) ::: CLOB);
    }
    float r;
    asm volatile("v_add_f32 %0, v16, v63" : "=v"(r) :: CLOB);
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MODE>
int run(float* d_out, int cus, const char* name)
{
    const int iters = 20000;
    for (int waves = 1; waves <= 8; waves *= 2) {
        const int blocks = cus * 4 * waves;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, iters);   // clock settle
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-46s waves/SIMD %d: %8.3f ms  ~%.2f clk per instruction per SIMD at 2.375 GHz\n", name, waves, ms, ms * 1e-3 * 2.375e9 / (24.0 * iters * waves));
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s %s CUs %d\n", p.name, p.gcnArchName, p.multiProcessorCount);
    float* d_out;
    CHECK(hipMalloc(&d_out, 4096));
    const int cus = p.multiProcessorCount;
    if (run<0>(d_out, cus, "pk_add  acc, c, acc   c in the other bank pair")) return 1;
    if (run<1>(d_out, cus, "pk_add  acc, c, acc   c in the same bank pair")) return 1;
    if (run<2>(d_out, cus, "pk_fma  acc, c, s, acc  c in the other pair")) return 1;
    if (run<3>(d_out, cus, "pk_fma  acc, c, s, acc  c in the same pair")) return 1;
    if (run<4>(d_out, cus, "pk_add  acc[j], c, acc[j+1] (ring shift)")) return 1;
    if (run<5>(d_out, cus, "pk_fma  acc, acc, s, acc")) return 1;
    return 0;
}
