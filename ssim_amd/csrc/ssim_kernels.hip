// ssim_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the SSIM hot path.
//
// Replaces, in one fused launch, the reference's per-tile chain
//   retrieve_tile x2 -> multiply x3 -> gaussian_blur x5 -> sum_tile     (src/ssim.cpp:747-783)
// and, in a second tiny launch, the final reduce (src/ssim.cpp:1090-1103, minus the division,
// which the host does in fp64 exactly as the reference: include/rmgr/ssim-hip.h finalize).
//
// Design (DESIGN.md has the long form):
//  * Work unit = one 64-lane wavefront = one workgroup = one vertical STRIP of the image:
//    128 (two per lane; ssim_strip2_kernel) or 64 (ssim_strip1_kernel) adjacent output columns x
//    strip_rows output rows.  No inter-wave communication, hence no real barriers; thousands of
//    strips fill 256 CUs.
//  * The strip walks DOWN its rows.  Per source row the wave (a) loads the row's uint8 pixels
//    of A and B (edge-clamped, any step/stride) one row ahead into registers, (b) converts them
//    and writes the five statistic planes a, b, a^2, b^2, ab of that row into a 2-slot LDS ring
//    (plane pairs (a,b) and (a^2,b^2) interleaved as float2 so that packed-fp32 VALU ops work
//    on naturally aligned register pairs), (c) every lane reads the 11(+1)-pixel window of its
//    columns back with wide ds_reads and runs the blur as a row SCATTER into an 11-deep register
//    ring of accumulators -- the reference's own formulation (src/ssim_fma.cpp:246-257), so that
//    MODE_EXACT reproduces its rounding order bit for bit -- and (d) the ring's oldest entry is
//    a finished output row: SSIM formula, fp64 accumulation, optional map store.
//  * Intermediates never touch HBM: algorithmic traffic is 2 B/pixel (+4 B/pixel with a map).
//  * The kernel is fp32-VALU bound (~140 packed instructions per pixel in MODE_EXACT), not HBM
//    bound; see DESIGN.md for the roofline arithmetic.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off   (fusion only where fma is written;
// float division stays IEEE correctly rounded -- div_inrange_* is the compiler's own sequence minus
// range handling that cannot trigger here -- and there is no -ffast-math anywhere).
#include "ssim_kernels.h"
#include <algorithm>
#include <cmath>
#include <type_traits>

// Cache policy of the map stores: nt (non-temporal).  The map is written once and never read by the kernel; as ordinary
// write-back lines it competes with the input rows for L2 and costs the separable mode 18 % with a map (8192^2: 333 -> 391 Gpix/s
// with nt; MODE_EXACT +2.7 %; profiles/r02_map_store_ab.txt).  Round 5 re-measured 0 (default) / 1 (sc0) / 2 (nt) / 3 (sc0 nt) /
// 18 (sc1 nt) on 2 x 8192^2: 369 / 371 / 395 / 392 / 395 Gpix/s separable, 196 / - / 200 / 200 / 199 exact: nt, any flavour
// (profiles/r05_prefetch_and_map_store_ab.txt).  What the map still costs -- 6...7 % in every mode against the same launch without
// it -- is neither its instructions (stores issued with a zero-record descriptor are free), nor HBM or the TLB (all rows folded onto
// eight cache-resident ones: the full cost), nor the pixel loads' latency (a second row in flight: +-0): it is proportional to the
// bytes that leave the CU (half the rows, or half the lanes: half the cost), and the float map is the contract.
#ifndef SSIM_MAP_STORE_AUX
#define SSIM_MAP_STORE_AUX 2
#endif

namespace ssim_hip {
namespace {

typedef float  f2 __attribute__((ext_vector_type(2)));
typedef float  f4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// Pointers that arrive inside structs lose their address space and would compile to flat_*
// accesses (which also tick the LDS counter).  Everything the kernel touches in memory is global.
typedef const uint8_t __attribute__((address_space(1)))*  gptr_u8;
typedef float __attribute__((address_space(1)))*          gptr_f32;
typedef double __attribute__((address_space(1)))*         gptr_f64;
typedef const PairDesc __attribute__((address_space(1)))* gptr_desc;

// ---------------------------------------------------------------------------------------------
// The 21 unique taps K(i,j), i<=j, of the reference's SIMD paths (src/ssim_fma.cpp:169-174):
// the float-computed 11x11 sigma=1.5 window.  Pure constexpr functions of literals so that
// every use folds to an inline constant / SGPR.
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr float ktab(int n)
{
    return n ==  0 ? 7.07622393965721130e-02f :
           n ==  1 ? 5.66619709134101868e-02f : n ==  2 ? 4.53713610768318176e-02f :
           n ==  3 ? 2.90912277996540070e-02f : n ==  4 ? 2.32944320887327194e-02f : n ==  5 ? 1.19597595185041428e-02f :
           n ==  6 ? 9.57662798464298248e-03f : n ==  7 ? 7.66836293041706085e-03f : n ==  8 ? 3.93706932663917542e-03f :
           n ==  9 ? 1.29605561960488558e-03f :
           n == 10 ? 2.02135881409049034e-03f : n == 11 ? 1.61857774946838617e-03f : n == 12 ? 8.31005279906094074e-04f :
           n == 13 ? 2.73561221547424793e-04f : n == 14 ? 5.77411265112459660e-05f :
           n == 15 ? 2.73561221547424793e-04f : n == 16 ? 2.19050692976452410e-04f : n == 17 ? 1.12464345875196159e-04f :
           n == 18 ? 3.70224843209143728e-05f : n == 19 ? 7.81441485742107034e-06f : 1.05756600987660931e-06f;
}
__host__ __device__ constexpr float kc(int i, int j) { return ktab(i <= j ? j * (j + 1) / 2 + i : i * (i + 1) / 2 + j); }

// ---------------------------------------------------------------------------------------------
// Small generic helpers over the stream value types: f2 (two planes or two columns packed ->
// v_pk_*_f32), float, d2, double.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float  fma_(float a, float b, float c)    { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ f2     fma_(f2 a, f2 b, f2 c)             { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ d2     fma_(d2 a, d2 b, d2 c)             { return __builtin_elementwise_fma(a, b, c); }

template <typename V> struct VT;
template <> struct VT<float>  { typedef float  S; static __device__ __forceinline__ float  splat(float k)  { return k; } };
template <> struct VT<f2>     { typedef float  S; static __device__ __forceinline__ f2     splat(float k)  { f2 v = {k, k}; return v; } };
template <> struct VT<double> { typedef double S; static __device__ __forceinline__ double splat(double k) { return k; } };
template <> struct VT<d2>     { typedef double S; static __device__ __forceinline__ d2     splat(double k) { d2 v = {k, k}; return v; } };

// Wave-uniform 64-bit value -> SGPR pair.
__device__ __forceinline__ int64_t uniform64(int64_t v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ double to_f64(float v) { return (double)v; }
__device__ __forceinline__ d2     to_f64(f2 v)    { d2 r = {(double)v.x, (double)v.y}; return r; }

// One multiply-accumulate step of the blur: fused (FMA path, src/ssim_fma.cpp:210-243) or with
// separately rounded product and sum (MUL_ADD of the generic/SSE/AVX paths, src/ssim.cpp:353).
template <bool FUSED, typename V>
__device__ __forceinline__ V mad(V a, V b, V c)
{
    if constexpr (FUSED) return fma_(a, b, c);
    else                 return c + a * b;
}

// sum_J = s0*K(0,J) then five multiply-adds in i order (src/ssim_fma.cpp:203-243).
template <int J, bool FUSED, typename V>
__device__ __forceinline__ V row_sum(V s0, V s1, V s2, V s3, V s4, V s5)
{
    V t = s0 * VT<V>::splat(kc(0, J));
    t = mad<FUSED>(s1, VT<V>::splat(kc(1, J)), t);
    t = mad<FUSED>(s2, VT<V>::splat(kc(2, J)), t);
    t = mad<FUSED>(s3, VT<V>::splat(kc(3, J)), t);
    t = mad<FUSED>(s4, VT<V>::splat(kc(4, J)), t);
    t = mad<FUSED>(s5, VT<V>::splat(kc(5, J)), t);
    return t;
}

// Reference-order blur of one source row into the 11-deep accumulator ring of one stream.
// acc[k] is the partial result of output row (r-5+k) while source row r is processed; the
// scatter "rows r-5..r+5 += sum5,sum4,..,sum0,..,sum5" (src/ssim_fma.cpp:246-257) becomes a
// shift of the ring (3-address adds: the shift is free).  Every output row therefore receives
// its 11 addends in source-row order, first one added to zero, exactly like the reference.
//
// KMIN (warm-up rows only, round 3): while the strip's first ten source rows -- the ones above its first output row -- pass
// through, ring entry k stands for an output row ABOVE the strip whenever k < 10 - i (i = 0..9 the warm-up row), and nothing
// ever reads it: the scatter into entries below KMIN is skipped, and with it every row sum S_j that only those entries use
// (the compiler drops them as dead).  Entries that will become outputs of the strip receive exactly what they always did.
template <bool FUSED, int KMIN = 0, typename V>
__device__ __forceinline__ void ring_scatter(V (&acc)[11], const V (&S)[6])
{
#pragma unroll
    for (int k = 0; k < 10; ++k)
        if (k >= KMIN) acc[k] = S[k < 5 ? 5 - k : k - 5] + acc[k + 1];
    acc[10] = S[5];
}
template <bool FUSED, int KMIN = 0, typename V>
__device__ __forceinline__ void blur_exact(V (&acc)[11], V s0, V s1, V s2, V s3, V s4, V s5)
{
    const V S[6] = {row_sum<0, FUSED>(s0, s1, s2, s3, s4, s5), row_sum<1, FUSED>(s0, s1, s2, s3, s4, s5), row_sum<2, FUSED>(s0, s1, s2, s3, s4, s5),
                    row_sum<3, FUSED>(s0, s1, s2, s3, s4, s5), row_sum<4, FUSED>(s0, s1, s2, s3, s4, s5), row_sum<5, FUSED>(s0, s1, s2, s3, s4, s5)};
    ring_scatter<FUSED, KMIN>(acc, S);
}

// acc[k] = h * g + acc[k+1]: the ring step of the separable blur.  In fp32 the packed FMA only exists in
// three-address form, which is what makes the ring shift free.  For fp64 the compiler prefers the two-address
// v_fmac_f64 (in place over acc[k+1]) and then copies every entry back into its loop-carried register -- one
// v_mov_b64 per FMA; spelling the three-address instruction out avoids that.  (Plain VALU-to-VALU dependency:
// interlocked by the hardware, nothing here for the hazard recogniser to miss.)
__device__ __forceinline__ f2 ring_fma(f2 h, float g, f2 c)   { return fma_(h, f2{g, g}, c); }
__device__ __forceinline__ float ring_fma(float h, float g, float c) { return fma_(h, g, c); }
__device__ __forceinline__ double ring_fma(double h, double g, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(h), "s"(g), "v"(c));
    return d;
}
__device__ __forceinline__ d2 ring_fma(d2 h, double g, d2 c)
{
    d2 d;
    d.x = ring_fma(h.x, g, c.x);
    d.y = ring_fma(h.y, g, c.y);
    return d;
}

// Separable blur (MODE_SEPARABLE and the mu streams of MODE_FAST in fp32 / MODE_DOUBLE fp64): 1-D pass along the row on the folded
// sums, then the vertical pass as the same ring scatter.  g[] = centre..edge taps of the true
// 1-D Gaussian (g(x)g(y) equals the 2-D kernel of tests/ssim_naive.h to 7e-18).
//
// ORDER: the order in which the row pass adds its six terms (tap index 0 = centre .. 5 = edge).  Every fp32 user now
// adds them centre first -- the reference's own order (src/ssim_fma.cpp:203-243), chosen without looking at any image.
// (Round 2 picked other orders per stream on the 18 small fixtures to steer the global value between two tolerances;
// on the reference's full-size sets that tuning did not hold -- DESIGN.md section 2 -- and it is gone.  The other two
// orders stay available to tests/tools/fast_mode_model.py's study of them.)
enum { ORDER_CENTRE_FIRST = 0, ORDER_SMALL_FIRST = 1, ORDER_INNER_FIRST = 2 };
template <int ORDER> __device__ constexpr int tap_at(int k)
{
    return ORDER == ORDER_CENTRE_FIRST ? k : ORDER == ORDER_SMALL_FIRST ? 5 - k : (k < 3 ? 2 - k : k);
}

template <int ORDER = ORDER_CENTRE_FIRST, int KMIN = 0, typename V, typename G>
__device__ __forceinline__ void blur_separable(V (&acc)[11], V s0, V s1, V s2, V s3, V s4, V s5, const G (&g)[6])
{
    const V s[6] = {s0, s1, s2, s3, s4, s5};
    V h = s[tap_at<ORDER>(0)] * VT<V>::splat(g[tap_at<ORDER>(0)]);
#pragma unroll
    for (int k = 1; k < 6; ++k)
        h = fma_(s[tap_at<ORDER>(k)], VT<V>::splat(g[tap_at<ORDER>(k)]), h);
#pragma unroll
    for (int k = 0; k < 10; ++k)
        if (k >= KMIN) acc[k] = ring_fma(h, g[k < 5 ? 5 - k : k - 5], acc[k + 1]);       // KMIN: see ring_scatter
    acc[10] = h * VT<V>::splat(g[5]);
}

// The same for two independent streams (the lane's two columns), their row passes interleaved: a stream's six
// multiply-adds form one dependent chain, and back-to-back dependent packed instructions cost a wait state each on
// gfx950 (the compiler pads them with s_nop) -- with two waves per SIMD there is nobody else to fill those slots.
template <int ORDER = ORDER_CENTRE_FIRST, typename V, typename G>
__device__ __forceinline__ void separable_rows_pair(V& hA, V& hB, const V (&a)[6], const V (&b)[6], const G (&g)[6])
{
    hA = a[tap_at<ORDER>(0)] * VT<V>::splat(g[tap_at<ORDER>(0)]);
    hB = b[tap_at<ORDER>(0)] * VT<V>::splat(g[tap_at<ORDER>(0)]);
#pragma unroll
    for (int k = 1; k < 6; ++k) {
        hA = fma_(a[tap_at<ORDER>(k)], VT<V>::splat(g[tap_at<ORDER>(k)]), hA);
        hB = fma_(b[tap_at<ORDER>(k)], VT<V>::splat(g[tap_at<ORDER>(k)]), hB);
    }
}
template <int KMIN = 0, typename V, typename G>
__device__ __forceinline__ void separable_columns_pair(V (&accA)[11], V (&accB)[11], V hA, V hB, const G (&g)[6])
{
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const int t = k < 5 ? 5 - k : k - 5;
        if (k >= KMIN) {                           // KMIN: see ring_scatter
            accA[k] = ring_fma(hA, g[t], accA[k + 1]);
            accB[k] = ring_fma(hB, g[t], accB[k + 1]);
        }
    }
    accA[10] = hA * VT<V>::splat(g[5]);
    accB[10] = hB * VT<V>::splat(g[5]);
}
template <int ORDER = ORDER_CENTRE_FIRST, int KMIN = 0, typename V, typename G>
__device__ __forceinline__ void blur_separable_pair(V (&accA)[11], V (&accB)[11], const V (&a)[6], const V (&b)[6], const G (&g)[6])
{
    V hA, hB;
    separable_rows_pair<ORDER>(hA, hB, a, b, g);
    separable_columns_pair<KMIN>(accA, accB, hA, hB, g);
}

// Per-pixel SSIM, unfused fp32 exactly as src/ssim.cpp:681-693 / src/ssim_avx.cpp:342-352.
// 2*x + c is written as fma(2,x,c): 2*x is exact, so the single rounding is the same one.
// EXACT_DIV = false (MODE_FAST since round 5; its contract is a tolerance against the FMA path, not its bits): the quotient as
// num * rcp(den), v_rcp_f32 being accurate to 1 ulp -- <= 1.8e-7 of a value in [-1, 1] against a per-pixel budget of 6.3e-4 --
// instead of the correctly rounded division the bit-exact modes need (6 packed instructions per pixel pair less, 2 % of the hybrid).
template <bool EXACT_DIV = true>
__device__ __forceinline__ float ssim_px(float muA, float muB, float eAA, float eBB, float eAB, float c1, float c2)
{
    const float muA2 = muA * muA, muB2 = muB * muB, muAB = muA * muB;
    const float sA2 = eAA - muA2, sB2 = eBB - muB2, sAB = eAB - muAB;
    const float num = __builtin_fmaf(2.0f, muAB, c1) * __builtin_fmaf(2.0f, sAB, c2);
    const float den = ((muA2 + muB2) + c1) * ((sA2 + sB2) + c2);
    if constexpr (EXACT_DIV) return num / den;
    else                     return num * __builtin_amdgcn_rcpf(den);
}
__device__ __forceinline__ double ssim_px(double muA, double muB, double eAA, double eBB, double eAB, double c1, double c2)
{
    const double muA2 = muA * muA, muB2 = muB * muB, muAB = muA * muB;
    const double sA2 = eAA - muA2, sB2 = eBB - muB2, sAB = eAB - muAB;
    const double num = (2 * muAB + c1) * (2 * sAB + c2);
    const double den = ((muA2 + muB2) + c1) * ((sA2 + sB2) + c2);
    return num / den;
}

// Correctly rounded n/d for operands that need none of v_div_scale / v_div_fixup: the instruction sequence the
// compiler emits for an IEEE float division, minus its range handling.  v_div_scale_f32 rescales only when an
// operand or the quotient is denormal, |exponent(n) - exponent(d)| >= 96, or d is within 2^3 of overflow, and
// v_div_fixup_f32 only replaces the result for zero / infinite / NaN operands (a zero numerator included, for
// which the sequence below also yields +0).  Here d = (muA^2+muB^2+c1)(sA^2+sB^2+c2) lies in [2^8, 2^35]
// (the variances carry at most ~0.1 of negative rounding noise against c2 = 58.5) and n =
// (2 muAB + c1)(2 sAB + c2) is 0 or has 2^-16 <= |n| <= 2^35 (2 sAB + c2 is a multiple of 2^-18 whenever it
// is small), so every intermediate below is the one the full sequence would produce: bit-identical quotients,
// two at a time in packed fp32.
__device__ __forceinline__ float opaque(float v) { asm("" : "+v"(v)); return v; }

// First half: the refined reciprocal of d; second half: the quotient and its two correction steps.
__device__ __forceinline__ f2 div_inrange_rcp(f2 d)
{
    const f2 one = {1.0f, 1.0f};
    const f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    return fma_(fma_(-d, r, one), r, r);
}
__device__ __forceinline__ f2 div_inrange_finish(f2 n, f2 d, f2 r)
{
    f2 q = n * r;
    q = fma_(fma_(-d, q, n), r, q);
    return fma_(fma_(-d, q, n), r, q);
}

// ssim_px for the two columns of a lane at once: (muA,muB) and (E[a^2],E[b^2]) arrive packed per column,
// E[ab] packed across the columns.  Same operations, same roundings as the scalar form above.  In two parts:
// everything that does not involve E[ab] -- the denominator and its refined reciprocal -- is issued together
// with the ab stream's blur (independent work to hide the dependent chain behind), the rest after it.
struct Px2 { f2 muAB, den, rcp; };
template <bool EXACT_DIV = true>
__device__ __forceinline__ Px2 ssim_px2_head(f2 mu0, f2 mu1, f2 e0, f2 e1, float c1, float c2)
{
    const f2 m0 = mu0 * mu0, m1 = mu1 * mu1;             // (muA^2, muB^2) per column
    const f2 s0 = e0 - m0, s1 = e1 - m1;                 // (sA^2, sB^2)
    // the cross-half terms stay scalar instructions (opaque(): keeps the vector combiner from turning them
    // into shuffles + packed ops); their results land in adjacent registers and continue packed
    Px2 h;
    h.muAB = f2{opaque(mu0.x * mu0.y), opaque(mu1.x * mu1.y)};
    const f2 C1 = {c1, c1}, C2 = {c2, c2};
    const f2 tm = {opaque(m0.x + m0.y), opaque(m1.x + m1.y)}, ts = {opaque(s0.x + s0.y), opaque(s1.x + s1.y)};
    h.den = (tm + C1) * (ts + C2);
    if constexpr (EXACT_DIV) h.rcp = div_inrange_rcp(h.den);
    else                     h.rcp = f2{__builtin_amdgcn_rcpf(h.den.x), __builtin_amdgcn_rcpf(h.den.y)};      // see ssim_px
    return h;
}
template <bool EXACT_DIV = true>
__device__ __forceinline__ f2 ssim_px2_tail(const Px2& h, f2 eAB, float c1, float c2)
{
    const f2 two = {2.0f, 2.0f}, C1 = {c1, c1}, C2 = {c2, c2};
    const f2 sAB = eAB - h.muAB;
    const f2 n = fma_(two, h.muAB, C1) * fma_(two, sAB, C2);
    if constexpr (EXACT_DIV) return div_inrange_finish(n, h.den, h.rcp);
    else                     return n * h.rcp;
}

// MODE_SEPARABLE and MODE_DOUBLE work on FOUR blurred planes, not five: the SSIM formula needs the two variances only as
// their sum, sigma_a^2 + sigma_b^2 = E[a^2 + b^2] - (mu_a^2 + mu_b^2), so a^2 + b^2 is blurred as one plane.  A fifth of
// the blur work, of the accumulator registers and of the staged LDS bytes goes away, and the ab plane can share a float2
// with it -- (a^2 + b^2, ab) per pixel -- which packs like the (a, b) plane does.
//
// Correctly rounded n/d in fp64 for the same operand ranges as div_inrange_* above (d in [2^8, 2^35], n = 0 or
// 2^-16 <= |n| <= 2^35: no scaling, no fix-up needed): an fp32 reciprocal seed (v_rcp_f32: 1 ulp of 2^-23), two Newton
// steps in fp64 (2^-46, then below 2^-53), the quotient and one residual correction.  The compiler's generic sequence
// for a double division is v_div_scale_f64 x2, the quarter-rate v_rcp_f64, eight v_fma_f64, v_div_fmas, v_div_fixup.
__device__ __forceinline__ double div_inrange(double n, double d)
{
    double r = (double)__builtin_amdgcn_rcpf((float)d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}
// One pixel of MODE_DOUBLE (uncentred moments: fp64 has the headroom).
__device__ __forceinline__ double ssim_px_four(double muA, double muB, double eS, double eX, double c1, double c2)
{
    const double muAB = muA * muB;
    const double tm = muA * muA + muB * muB;       // mul, mul, add: three roundings (-ffp-contract=off)
    const double ts = eS - tm, sAB = eX - muAB;
    const double num = __builtin_fma(2.0, muAB, c1) * __builtin_fma(2.0, sAB, c2);     // 2x is exact: the one rounding of 2x + c
    const double den = (tm + c1) * (ts + c2);
    return div_inrange(num, den);
}

// MODE_SEPARABLE blurs CENTRED pixels a' = a - 128, b' = b - 128 (exact in fp32): variance and covariance do not move
// with the origin, sigma_a^2 + sigma_b^2 = E[a'^2 + b'^2] - (mu_a'^2 + mu_b'^2), sigma_ab = E[a'b'] - mu_a' mu_b', but the
// cancellation that dominates fp32 SSIM error -- E[x^2] - mu^2 with both near 65025 in bright flat areas, against
// c2 = 58.5 -- now happens between numbers of at most 16384 (32768 for the sum plane): the worst per-pixel error
// against the exact value drops from 6.9e-4 to 1.9e-4 on the reference's test sets (tests/tools/fast_mode_model.py) for
// two subtractions per staged pixel and a handful of operations per output pixel.  The luminance term uses mu = mu' + 128.
// Scalar form (the one-column kernel); the packed form below performs the same operations on both columns of a lane, so
// the two kernels agree bit for bit.
constexpr float kCentre = 128.0f;
// Round 5: n / d as n * rcp(d) in this mode.  It is not bit-exact against anything (its contract: the reference's TEST tolerances
// against the exact value, include/rmgr/ssim-hip.h); v_rcp_f32 is accurate to 1 ulp and the product adds half an ulp: <= 1.8e-7 of a
// value in [-1, 1], three orders of magnitude inside the mode's per-pixel budget, for 3 instead of 9 instructions per two pixels
// (the correctly rounded sequence of div_inrange_*: one Newton step on the reciprocal, two quotient corrections).  d is in
// [2^8, 2^35] as before (ssim_px2_head): no scaling, no special cases.  What was tried with it and dropped (tests/tools/
// fast_mode_model.py, profiles/r05_separable_epilogue.md): expanding the luminance terms around the centre
// (2 mu_a mu_b + c1 = 2 m_a m_b + 256 (m_a + m_b) + 2 * 128^2 + c1, two packed and two scalar operations fewer) cancels
// catastrophically in dark areas -- 4.1e-4 per pixel on bbb360 against 1.5e-4 --, and mu_a^2 + mu_b^2 = (m_a - m_b)^2 + 2 m_a m_b
// keeps the per-pixel error but biases the global value (1.4e-6 on the bbb crops against 8.7e-7).
// Scalar form (the one-column kernel); the packed form below performs the same operations on both columns of a lane, so
// the two kernels agree bit for bit.
__device__ __forceinline__ float ssim_px_sep(float mA, float mB, float eS, float eX, float c1, float c2)
{
    const float sS = eS - (mA * mA + mB * mB), sAB = eX - mA * mB;       // from the centred moments
    const float muA = mA + kCentre, muB = mB + kCentre;
    const float muAB = muA * muB, tm = muA * muA + muB * muB;
    const float n = __builtin_fmaf(2.0f, muAB, c1) * __builtin_fmaf(2.0f, sAB, c2);
    const float d = (tm + c1) * (sS + c2);
    return n * __builtin_amdgcn_rcpf(d);
}
// m0, m1 = centred (mu_a', mu_b') of the lane's two columns; e0, e1 = (E[a'^2 + b'^2], E[a'b']) of the two columns
__device__ __forceinline__ f2 ssim_px2_sep(f2 m0, f2 m1, f2 e0, f2 e1, float c1, float c2)
{
    const f2 q0 = m0 * m0, q1 = m1 * m1;
    const f2 pc = {opaque(m0.x * m0.y), opaque(m1.x * m1.y)};            // mu_a' mu_b'
    const f2 tc = {opaque(q0.x + q0.y), opaque(q1.x + q1.y)};            // mu_a'^2 + mu_b'^2
    const f2 sS = {opaque(e0.x - tc.x), opaque(e1.x - tc.y)};
    const f2 sAB = {opaque(e0.y - pc.x), opaque(e1.y - pc.y)};
    const f2 off = {kCentre, kCentre};
    const f2 u0 = m0 + off, u1 = m1 + off;                               // (mu_a, mu_b)
    const f2 v0 = u0 * u0, v1 = u1 * u1;
    const f2 muAB = {opaque(u0.x * u0.y), opaque(u1.x * u1.y)};
    const f2 tm = {opaque(v0.x + v0.y), opaque(v1.x + v1.y)};
    const f2 two = {2.0f, 2.0f}, C1 = {c1, c1}, C2 = {c2, c2};
    const f2 n = fma_(two, muAB, C1) * fma_(two, sAB, C2);
    const f2 den = (tm + C1) * (sS + C2);
    const f2 r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    return n * r;
}

struct KArgs {
    PairDesc        single;       // used when descs == nullptr
    const PairDesc* descs;
    uint32_t        width, height, strip_rows, strips_x, strips_y;
    uint32_t        y_begin, y_end;   // output rows of this launch (a multiple-of-8 start; the whole image unless the host pipelines bands)
    uint32_t        cells_x, cells_y; // the image's grid of 64-column x cell_rows-row cells: the units of the fp64 reduction
    uint32_t        cell_shift;       // log2(cell_rows): 3 or 5 by image height (cell_rows_for())
    uint32_t        col_cells;    // cell rows of one strip column inside [y_begin, y_end)
    uint32_t        chunk_cells;  // 0: one strip per workgroup (grid strips_x x strips_y x count); > 0: the BALANCED schedule -- workgroup w covers cell rows
                                  // [w * chunk_cells, ...) of the launch's flattened [image][strip column][cell row] list, continuing into the next column / image
    uint32_t        n_chunks;     // workgroups of the balanced schedule (1-D grid)
    uint32_t        bal_stride;   // balanced schedule: T >= 1 images are interleaved column by column in the flattened list (list position -> image, column: list_column())
    uint32_t        xcds;         // XCDs the dispatcher deals consecutive workgroups to (hipDeviceAttributeNumberOfXccs; 8 on MI355X)
    uint32_t        count;        // images in this launch == gridDim.z (reading gridDim itself is a fetch from the dispatch packet)
    uint32_t        group;        // >1: runs of `group` consecutive descriptors address interleaved channels of one image pair
    double*         partials;     // [image][cell_y][cell_x]
    uint64_t*       clock;        // profiling only (NULL otherwise): the first `xcds` workgroups add the shader cycles (s_memtime) and the reference ticks (s_memrealtime) of their
                                  // own run to per-XCD counters -- the shader clock the kernel really ran at on every XCD (clock_begin / clock_end)
    float           c1, c2;
    float           gf[6];        // separable taps, fp32
    double          c1d, c2d;
    double          gd[6];        // separable taps, fp64
};

// One workgroup is one wavefront, and a wave's LDS operations execute in issue order, so lanes only need
// the COMPILER to keep LDS accesses in program order across this point -- no s_barrier, and above all no
// s_waitcnt: __syncthreads() makes the wave drain its just-issued staging writes (~150 cycles per row).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// What every strip kernel does first: find its strip and its image pair.
struct Strip {
    PairDesc pd;                 // wave-uniform (SGPRs)
    int64_t  W, H, x0, y0, y_end;
    uint32_t sx, sy, img;
};

// Profiling aid (rmgr_ssim_hip_set_profiling): the shader clock a launch really ran at, per XCD.  The first `xcds` workgroups of a launch -- one per XCD: the dispatcher
// deals consecutive workgroup ids to the XCDs round robin -- note s_memtime (one tick per shader cycle) and s_memrealtime (the constant reference clock, 100 MHz) when
// they start and add the differences to device counters when they end: frequency of XCD x = reference rate x cycles[x] / ticks[x], accumulated over the profiled
// launches.  Layout: CLOCK_STRIDE uint64 per XCD: start cycles, start ticks, sum of cycles, sum of ticks, launches.  Nothing is held in registers in between (the start
// values wait in memory), and with clock == NULL it is one scalar branch at either end of the kernel.
enum { CLOCK_STRIDE = 5 };
__device__ __forceinline__ void clock_begin(uint64_t* clock)
{
    const uint64_t t = __builtin_readcyclecounter(), r = __builtin_amdgcn_s_memrealtime();
    uint64_t* c = clock + CLOCK_STRIDE * blockIdx.x;
    if (threadIdx.x == 0) { c[0] = t; c[1] = r; }
}
__device__ __forceinline__ void clock_end(uint64_t* clock)
{
    const uint64_t t = __builtin_readcyclecounter(), r = __builtin_amdgcn_s_memrealtime();
    uint64_t* c = clock + CLOCK_STRIDE * blockIdx.x;
    if (threadIdx.x == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(c + 2), (unsigned long long)(t - c[0]));
        atomicAdd(reinterpret_cast<unsigned long long*>(c + 3), (unsigned long long)(r - c[1]));
        atomicAdd(reinterpret_cast<unsigned long long*>(c + 4), 1ull);
    }
}
#define SSIM_CLOCKED(args) ((args).clock != nullptr && blockIdx.x < (args).xcds && blockIdx.y == 0 && blockIdx.z == 0)

// Workgroup g of `total` -> its place in the work list when each of the `xcds` XCDs (which the dispatcher deals consecutive
// workgroup ids to, round robin) is to walk ONE contiguous share of the list: XCD k owns total / xcds entries, the first
// total % xcds XCDs one more.  A bijection for any total and any XCD count; wave-uniform scalar arithmetic.
__device__ __forceinline__ uint32_t xcd_order(uint32_t g, uint32_t total, uint32_t xcds)
{
    uint32_t xcd, slot, q, rem;
    if (xcds == 8) { xcd = g & 7u; slot = g >> 3; q = total >> 3; rem = total & 7u; }      // MI355X: no division
    else           { slot = g / xcds; xcd = g - slot * xcds; q = total / xcds; rem = total - q * xcds; }
    return xcd * q + (xcd < rem ? xcd : rem) + slot;
}

__device__ __forceinline__ Strip strip_setup(const KArgs& args, int strip_w)
{
    Strip st;
    // XCD-aware strip order.  The dispatcher hands consecutive workgroup ids (x fastest, then y, then z) to the
    // 8 XCDs round robin; left to itself that puts horizontally adjacent strips -- which share their 2 x 8 halo
    // columns and hence cache lines -- on different L2s (measured: 3.07x the algorithmic HBM reads).  Renumber
    // so that each XCD walks one contiguous eighth of the batch's strip list (image-major, then strip row, then
    // strip column): 1.07x.  A bijection for any strip count; speed only, correctness does not depend on it.
    // args.group > 1: the batch consists of runs of `group` sibling images that live in the same bytes (the
    // channels of an interleaved pixel format); their strips at one position then run back to back on one XCD
    // and share its L2 lines (position-major, sibling-minor walk) instead of each channel re-fetching them.
    const uint32_t per_img = args.strips_x * args.strips_y;
    if (args.xcds == 8 && (per_img & 7u) == 0 && args.group <= 1) {
        // the common case needs no division by per_img: every image's strip list splits into eighths
        const uint32_t lin = blockIdx.y * args.strips_x + blockIdx.x;
        const uint32_t swz = (lin & 7u) * (per_img >> 3) + (lin >> 3);
        st.img = blockIdx.z;
        st.sy = swz / args.strips_x;
        st.sx = swz - st.sy * args.strips_x;
    } else {
        const uint32_t total = per_img * args.count;
        const uint32_t g = (blockIdx.z * args.strips_y + blockIdx.y) * args.strips_x + blockIdx.x;
        const uint32_t id = xcd_order(g, total, args.xcds);
        uint32_t lin;
        if (args.group > 1) {
            const uint32_t span = per_img * args.group;
            const uint32_t run = id / span, within = id - run * span;
            lin = within / args.group;
            st.img = run * args.group + (within - lin * args.group);
        } else {
            st.img = id / per_img;
            lin = id - st.img * per_img;
        }
        st.sy = lin / args.strips_x;
        st.sx = lin - st.sy * args.strips_x;
    }
    // Everything in the descriptor is wave-uniform: keep it in SGPRs so that row addressing is
    // scalar arithmetic (a descriptor fetched through a pointer would otherwise sit in VGPRs).
    st.pd = args.single;
    if (args.descs) {
        const gptr_desc gd = (gptr_desc)args.descs + st.img;
        st.pd.a = (const uint8_t*)uniform64((int64_t)gd->a); st.pd.a_step = uniform64(gd->a_step); st.pd.a_stride = uniform64(gd->a_stride);
        st.pd.b = (const uint8_t*)uniform64((int64_t)gd->b); st.pd.b_step = uniform64(gd->b_step); st.pd.b_stride = uniform64(gd->b_stride);
        st.pd.map = (float*)uniform64((int64_t)gd->map); st.pd.map_step = uniform64(gd->map_step); st.pd.map_stride = uniform64(gd->map_stride);
    }
    st.W = args.width; st.H = args.height;
    st.x0 = (int64_t)st.sx * strip_w;
    st.y0 = (int64_t)args.y_begin + (int64_t)st.sy * args.strip_rows;
    st.y_end = (st.y0 + args.strip_rows < (int64_t)args.y_end) ? st.y0 + args.strip_rows : (int64_t)args.y_end;
    return st;
}

// BALANCED schedule of the two-column kernel (round 5; prototyped in round 4: profiles/r04_balanced_schedule_ab.txt).  plan()'s strips
// quantise: 128 x 1080p are 1920 strip columns x 1080 rows on 2048 wave slots; the best split, three 360-row strips per column, is
// 5760 strips = 2.81 rounds of 372 row-times = 1116 against 1012.5 if the work were spread evenly.  Here the launch's cell rows are
// flattened [image][strip column][cell row] and cut into `wave slots` equal chunks; a wavefront walks its chunk SEGMENT by segment
// (a segment = the part inside one strip column: what a strip is), continuing into the next strip column or image.  Cells, maps
// and sums are the strips' bit for bit.  It pays where the strips leave a partial round and LOSES elsewhere (a static equal
// partition needs every SIMD to run at the same speed for the whole launch): which launches take it is plan()'s measured rule.
// Round 6: the PHASE of the chunks.  A wavefront reads three 128-byte lines per source row and image -- its own and a few bytes of each
// horizontal neighbour's -- and the XCD's 4 MiB L2 holds ~20 rows of its 256 wavefronts: two of the three lines hit only if the neighbouring
// strip columns of an image are walked within ~20 rows of each other.  The strips are (all start together at the same rows).  Equal chunks of
// c cell rows cut out of columns of C are not: chunk k starts at cell row k c mod C of its column, so neighbouring columns are C mod c cell
// rows out of step (128 x 1080p, c = 127, C = 135: 64 rows) and round 5's launch fetched 2.9...3.0x its algorithmic bytes where the strips fetch
// 1.02...1.12x -- and ran up to 2.3 % slower for it (profiles/r06_phase_ab.txt).  Remedy: T images are INTERLEAVED column by column in the
// list (list positions k T + i hold column k of the block's image i), with T chosen by plan() so that T (C mod c) is within a cell row or so of
// a multiple of c: neighbouring columns of one image, T list positions apart, are then that little out of step instead of C mod c.
// 128 x 1080p (T = 16): 2.90x -> 1.04x, +2.3 %; 96 x 1080p (T = 19) +2.7 %; 64 x 1080p (T = 9) 2.94x -> 1.24x; 32 x 1080p 3.01x -> 1.53x (what is
// left there: the columns at the ends of an XCD's share, whose neighbours another XCD walks).  Also measured: a chunk's two segments walked
// last to first (every wavefront starts a column at its row 0): 128 x 1080p 1.12x, +1.3 % -- the interleave is better everywhere and is what ships.
// Same segments, same cells, same sums: only which wavefront takes which segment.
// List position `col` of the flattened [block of T images][strip column][image of the block] list -> image and strip column.
__device__ __forceinline__ void list_column(const KArgs& args, uint32_t col, uint32_t& img, uint32_t& sx)
{
    const uint32_t T = args.bal_stride;
    if (T <= 1) {
        img = col / args.strips_x;
        sx = col - img * args.strips_x;
        return;
    }
    const uint32_t span = T * args.strips_x, blk = col / span, within = col - blk * span;
    const uint32_t left = args.count - blk * T, here = left < T ? left : T;     // the last block may hold fewer images
    sx = within / here;
    img = blk * T + (within - sx * here);
}

// The work of one wavefront: cell rows [first, end) of the flattened list; `first` moves up as segments are taken.
struct Work { uint32_t first, end; };

__device__ __forceinline__ Work work_setup(const KArgs& args)
{
    Work w;
    // each XCD walks a contiguous share of the chunk list (strip_setup)
    const uint32_t id = xcd_order(blockIdx.x, args.n_chunks, args.xcds);
    const uint32_t all = args.count * args.strips_x * args.col_cells;      // < 2^31 (plan())
    w.first = id * args.chunk_cells;
    w.end = all - w.first < args.chunk_cells ? all : w.first + args.chunk_cells;
    return w;
}

// The next segment of `w` as a Strip; shrinks `w`.
__device__ __forceinline__ Strip segment_setup(const KArgs& args, Work& w, int strip_w)
{
    Strip st;
    const uint32_t col = w.first / args.col_cells, cy = w.first - col * args.col_cells;
    const uint32_t seg = args.col_cells - cy < w.end - w.first ? args.col_cells - cy : w.end - w.first;
    w.first += seg;
    list_column(args, col, st.img, st.sx);
    st.sy = 0;
    st.pd = args.single;
    if (args.descs) {
        const gptr_desc gd = (gptr_desc)args.descs + st.img;
        st.pd.a = (const uint8_t*)uniform64((int64_t)gd->a); st.pd.a_step = uniform64(gd->a_step); st.pd.a_stride = uniform64(gd->a_stride);
        st.pd.b = (const uint8_t*)uniform64((int64_t)gd->b); st.pd.b_step = uniform64(gd->b_step); st.pd.b_stride = uniform64(gd->b_stride);
        st.pd.map = (float*)uniform64((int64_t)gd->map); st.pd.map_step = uniform64(gd->map_step); st.pd.map_stride = uniform64(gd->map_stride);
    }
    st.W = args.width; st.H = args.height;
    st.x0 = (int64_t)st.sx * strip_w;
    st.y0 = (int64_t)args.y_begin + ((int64_t)cy << args.cell_shift);
    const int64_t ye = st.y0 + ((int64_t)seg << args.cell_shift);
    st.y_end = ye < (int64_t)args.y_end ? ye : (int64_t)args.y_end;
    return st;
}

// The fp64 reduction is organised in CELLS of 64 columns x cell_rows rows at absolute image positions, not in
// strips (cell_rows is 8 or 32, a function of the image height alone: cell_rows_for()).  Whatever the strip
// height, the kernel variant, the batch size or the row window of the launch, a cell's value is
//   (1) per column, the cell's (up to cell_rows) per-pixel values added in row order;
//   (2) the two columns of each even/odd pair added;
//   (3) a butterfly over the 32 pairs (partner distance 1, 2, 4, 8, 16; fp addition is commutative, so every
//       lane of the butterfly holds the same bits);
// and the per-image sum is ssim_reduce_kernel's fixed-order sum of the cells.  The sums are therefore bit-identical
// however a batch is cut into launches, strips, bands or GPUs.  Strips start on cell boundaries (plan()).  8 B of HBM per cell.
// Cost (round 3): cells are reduced EIGHT AT A TIME.  Round 2 ran the tree of every cell on its own, as four DPP levels + a
// readlane level on 64-bit values: ~80 issue slots per cell and wave, 2.4 % of MODE_EXACT on 1080p batches (8-row cells).
// Now a lane parks its leaf of cell k of the batch in LDS (one ds_write_b64 per cell) and, when eight cells are parked -- or
// the strip ends --, the wave reduces all of them at once: the batch is 8 cells x 64 leaves = 512 doubles, lane L reads the
// eight consecutive leaves L*8 .. L*8+7 (one eighth of a 64-leaf tree, or one quarter of a 32-leaf one), adds them as the
// balanced tree ((x0+x1)+(x2+x3))+((x4+x5)+(x6+x7)), and two or three DPP levels combine the lanes that share a tree --
// the same additions in the same association as the butterfly described above (its level at distance d adds neighbouring
// blocks of d leaves, and fp addition is commutative), so the cell values, and with them every sum, keep the bits they had:
// ~35 issue slots per EIGHT cells.  4 KiB of LDS per wave.
enum { CELL_BATCH = 8 };
struct CellBatch { double leaf[CELL_BATCH][64]; };

#define SSIM_DPP_ADD(t, CTRL) do {                                                                            \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(t), (CTRL), 0xF, 0xF, false);           \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(t), (CTRL), 0xF, 0xF, false);           \
        (t) += __hiloint2double(hi_, lo_);                                                                    \
    } while (0)
enum { DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141 };

// The eight leaves of this lane's share of the batch, reduced as the balanced tree ((x0+x1)+(x2+x3))+((x4+x5)+(x6+x7)),
// depth first and with the reads kept where they are used (fences): the kernels that run three waves per SIMD have
// no registers to spare for eight doubles on top of their accumulator rings.
__device__ __forceinline__ double cell_batch_local(const CellBatch& cb, int lane)
{
    const d2* p = reinterpret_cast<const d2*>(&cb.leaf[0][0]) + 4 * lane;       // leaves lane*8 .. lane*8+7
    d2 v = p[0];
    double a = v.x + v.y;
    __builtin_amdgcn_sched_barrier(0);
    v = p[1];
    a = a + (v.x + v.y);
    __builtin_amdgcn_sched_barrier(0);
    v = p[2];
    double b = v.x + v.y;
    __builtin_amdgcn_sched_barrier(0);
    v = p[3];
    b = b + (v.x + v.y);
    return a + b;
}

// two columns per lane: a leaf is the lane's column pair (column 2l + column 2l+1); leaves 0-31 of a batch row are cell
// 2*sx, leaves 32-63 cell 2*sx+1.  Four lanes share a 32-leaf tree.  `count` cells of the batch are real.
__device__ __forceinline__ void cell_batch_flush2(const KArgs& args, const Strip& st, const CellBatch& cb, uint32_t cell_y_first, uint32_t count)
{
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));             // everything derived from the lane id below is computed HERE, per flush: hoisted out
                                               // of the cell loop it would occupy registers next to the accumulator rings
    double t = cell_batch_local(cb, lane);
    SSIM_DPP_ADD(t, DPP_QUAD_XOR1);            // blocks of 8 leaves -> 16
    SSIM_DPP_ADD(t, DPP_QUAD_XOR2);            // 16 -> 32: the cell
    const uint32_t c = (uint32_t)lane >> 3, cx = 2u * st.sx + (((uint32_t)lane >> 2) & 1u);
    if ((lane & 3) == 0 && c < count && cx < args.cells_x)
        ((gptr_f64)args.partials)[((size_t)st.img * args.cells_y + cell_y_first + c) * args.cells_x + cx] = t;
}
// one column per lane: a leaf is the lane's column, the wave is one cell wide.  Eight lanes share a 64-leaf tree
// (whose first level adds the even/odd column pairs, as the two-column kernel does in registers).
__device__ __forceinline__ void cell_batch_flush1(const KArgs& args, const Strip& st, const CellBatch& cb, uint32_t cell_y_first, uint32_t count)
{
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));             // see cell_batch_flush2
    double t = cell_batch_local(cb, lane);
    SSIM_DPP_ADD(t, DPP_QUAD_XOR1);            // blocks of 8 leaves -> 16
    SSIM_DPP_ADD(t, DPP_QUAD_XOR2);            // 16 -> 32
    SSIM_DPP_ADD(t, DPP_ROW_HALF_MIRROR);      // 32 -> 64 (quads are uniform: lane i <-> 7-i of each 8 is a lane of the other quad)
    const uint32_t c = (uint32_t)lane >> 3;
    if ((lane & 7) == 0 && c < count)
        ((gptr_f64)args.partials)[((size_t)st.img * args.cells_y + cell_y_first + c) * args.cells_x + st.sx] = t;
}

// ---------------------------------------------------------------------------------------------
// ssim_strip2_kernel -- the default kernel of the fp32 modes: two adjacent columns per lane,
// 128-column strips.  LDS slot = one source row of 144 pixels starting SEVEN columns left of the strip
// (round 5; 5 are needed, rounds 1-4 had 8): the twelve window pixels of a lane's two columns -- columns
// x-5 .. x+6, slot pixels 2 lane + 2 .. 2 lane + 13 -- then start on an even slot pixel and are exactly six
// naturally aligned 16-byte reads per plane; with an even left halo they straddle seven, the first and last
// half unused (20 -> 18 window reads per row in the five-plane modes, 14 -> 12 in MODE_SEPARABLE, and 4
// window registers fewer per plane).  Planes: (a,b) pairs, (a*a,b*b) pairs, and the ab plane stored as
// pairs xx[p] = (ab[p], ab[p+1]) so that ALL five blur streams are packed-fp32 code on naturally
// aligned register pairs (no v_pk_mov shuffles), fed by 16-byte ds_read_b128.
// Tried and dropped (measured on MI355X, round 1): refilling each plane's window for row r+1 right
// after its last use (+36 VGPRs -> 1 wave/SIMD), skipping the row sums halo rows cannot use (branches
// in the hot loop: -10 %), forcing 3 waves/SIMD with launch bounds (spills: 5x slower), a second pixel
// row in flight from HBM (+-0), reordering the ring scatter so that the newest entry needs no copy (the
// copy moves to the oldest entry instead; unrolling over the two LDS slots is what removes it).
// ---------------------------------------------------------------------------------------------
struct Slot2 {
    static constexpr int STRIP_W = 128, PAD = 7, ROW_PX = 144;      // PAD: halo columns left of the strip (odd: see above); 137 + 5 pixels are used
    f2 ab[ROW_PX];   // (a, b)
    f2 q[ROW_PX];    // (a*a, b*b)
    f2 xx[ROW_PX];   // (ab[p], ab[p+1])
};

enum { ROW_WARMUP = 0, ROW_MAIN = 1, ROW_LAST = 2 };

// MODE_FAST (round 3): the HYBRID.  The three E[.] streams -- (a^2,b^2) of each column and ab -- run the reference's
// exact operation order (blur_exact) and are bit-identical to the reference's planes; only the two (a,b) streams, i.e.
// mu_a and mu_b, use the separable blur (centre-first row pass, fused column pass).  Why this split: on the reference's
// own test sets the FMA path's per-pixel error against the exact value is almost entirely the rounding of the E[.]
// planes (reference mu + exact E: 6.5e-4 from the reference; exact mu + reference E: 2.2e-4), and on bbb1080 that error
// (6.46e-4) is larger than north_star's FMA-relative tolerance (6.3e-4) -- so an implementation can only be inside
// the tolerance by REPRODUCING those roundings, and the mu planes are where it can afford not to.  51 -> 22 lane-ops
// per pixel on two of the five planes: 278 -> 220 per pixel.  Epilogue, LDS layout, rings: MODE_EXACT's.
// MODE_SEPARABLE (round 2's MODE_FAST, now on centred pixels): four planes -- the (a',b') pair plane and the
// (a'^2 + b'^2, a'b') pair plane (ssim_px2_sep above), both float2 per pixel, so the lane's four packed streams are
// (a',b') and (a'^2+b'^2, a'b') of each of its two columns; the `q` member of the slot holds the second plane, `xx`
// is not used.  14 window reads per lane-row (round 1: 20).
// Measured before that, on the five-plane form (profiles/r02_fast_lds_attack.md, r02_fast_interleave_ab.txt):
// products formed in registers instead of staged +3.3...4.4 %, interleaved row passes +1.4 %.
// MAP: 0 no map; 1 map with any ssimStep (one 4-byte store per column); 2 every pair of the launch has ssimStep == 1
// and the width is even (no lane owns a lone last column): the lane's two adjacent values go out as one 8-byte store
// (a wave writes 512 contiguous bytes per row; +2...3 % for MODE_SEPARABLE with a map, neutral for MODE_EXACT).
// EARLY (bit-exact modes only; round 3): the six row sums of the two (a,b) streams -- 72 packed instructions -- are formed with
// the fold at the bottom of the previous iteration, in the wave's low-priority phase, and cross the loop edge instead of
// the folded sums (24 registers either way); the high-priority phase then only scatters them into the ring.  Same
// operations, same bits.  It pays on launches of few rounds of wave slots and costs on long ones (see launch()).
// BAL (no map; bit-exact modes with EARLY, and MODE_FAST): the balanced schedule (work_setup above) -- the body below in a segment loop.
template <int MODE, int MAP, bool EARLY = false, bool BAL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MODE == MODE_SEPARABLE ? 3 : 2, MODE == MODE_SEPARABLE ? 3 : 2)))
void ssim_strip2_kernel(const KArgs args)
{
    constexpr bool FAST = (MODE == MODE_SEPARABLE);  // the four-plane, three-waves-per-SIMD flow of the row loop
    constexpr bool HYB = (MODE == MODE_FAST);        // MODE_EXACT's flow with separable (a,b) streams
    constexpr int PAD = Slot2::PAD, ROW_PX = Slot2::ROW_PX;
    constexpr int NLOAD = 3;                         // pixels each lane stages per row
    constexpr bool FUSED = (MODE != MODE_UNFUSED);
    static_assert(MODE != MODE_DOUBLE, "fp64 mode uses ssim_strip1_kernel");

    __shared__ __attribute__((aligned(16))) Slot2 ring[2];
    __shared__ __attribute__((aligned(16))) CellBatch cells;

    const int lane = threadIdx.x;
    // the separable taps as six scalars (an array inside the kernel argument block, passed on by reference, can end up
    // in scratch memory when scalar and packed streams both use it)
    const float gf[6] = {args.gf[0], args.gf[1], args.gf[2], args.gf[3], args.gf[4], args.gf[5]};
    if (SSIM_CLOCKED(args)) clock_begin(args.clock);
    Work wk = {0, 0};
    if constexpr (BAL) wk = work_setup(args);
#pragma unroll 1
    do {                                // BAL: one pass per segment of the wavefront's chunk; otherwise once: the wavefront's strip
    Strip st;
    if constexpr (BAL) st = segment_setup(args, wk, Slot2::STRIP_W);
    else               st = strip_setup(args, Slot2::STRIP_W);
    const PairDesc& pd = st.pd;
    // 32-bit coordinates (fits_strip2() on the host guarantees the ranges): row bookkeeping stays on the
    // scalar unit -- there are no 64-bit scalar ordered compares, so int64 loop counters cost VALU work.
    const int W = (int)st.W, H = (int)st.H, x0 = (int)st.x0, y0 = (int)st.y0, y_end = (int)st.y_end;

    // Per-lane staging columns: pixel p of the slot is image column clamp(x0 - PAD + p)
    // (edge replication of the IMAGE, src/ssim.cpp:529-554).  Addresses are a wave-uniform row base
    // (SGPR pair) plus a non-negative 32-bit lane offset from the strip's lowest-addressed column, the
    // global_load "saddr" form: nothing but the load itself is issued per row and per pixel.
    auto clampx = [&](int x) { return x < 0 ? 0 : (x > W - 1 ? W - 1 : x); };
    const int x_lo = clampx(x0 - PAD), x_hi = clampx(x0 - PAD + ROW_PX - 1);
    const int refA = pd.a_step >= 0 ? x_lo : x_hi, refB = pd.b_step >= 0 ? x_lo : x_hi;
    const gptr_u8 baseA = (gptr_u8)pd.a + (int64_t)refA * pd.a_step;
    const gptr_u8 baseB = (gptr_u8)pd.b + (int64_t)refB * pd.b_step;
    int      sp[NLOAD];
    uint32_t offA[NLOAD], offB[NLOAD];
#pragma unroll
    for (int t = 0; t < NLOAD; ++t) {
        int p = lane + 64 * t;
        p = p < ROW_PX ? p : ROW_PX - 1;
        const int xg = clampx(x0 - PAD + p);
        sp[t] = p;
        offA[t] = (uint32_t)((int64_t)(xg - refA) * pd.a_step);
        offB[t] = (uint32_t)((int64_t)(xg - refB) * pd.b_step);
    }

    // One row of pixels is in flight in registers (requested an iteration, ~1 us, before it is staged).
    // A second row in flight was measured twice: round 1 (no gain in MODE_EXACT, -3 % in the separable mode: registers) and round 5
    // (registers to spare by then; +-0 with and without the map in every mode: profiles/r05_prefetch_and_map_store_ab.txt).
    uint8_t va[NLOAD], vb[NLOAD];
    auto fetch_to = [&](int r, uint8_t (&oa)[NLOAD], uint8_t (&ob)[NLOAD]) {  // row r (clamped: src/ssim.cpp:562-582) -> registers
        const int ry = r < 0 ? 0 : (r > H - 1 ? H - 1 : r);
        const gptr_u8 ra = baseA + (int64_t)ry * pd.a_stride;
        const gptr_u8 rb = baseB + (int64_t)ry * pd.b_stride;
#pragma unroll
        for (int t = 0; t < NLOAD; ++t) {
            // The empty asm keeps the zero-extension of the offset inside the loop body, where instruction
            // selection can fold it into the load (hoisted, it becomes a 64-bit VGPR pair and a 64-bit add).
            asm volatile("" : "+v"(offA[t]), "+v"(offB[t]));
            oa[t] = ra[offA[t]];
            ob[t] = rb[offB[t]];
        }
    };
    auto fetch = [&](int r) { fetch_to(r, va, vb); };
    auto stage_from = [&](Slot2& s, const uint8_t (&ia)[NLOAD], const uint8_t (&ib)[NLOAD]) {    // registers -> the five planes of one LDS slot
        float* xf = reinterpret_cast<float*>(s.xx);
#pragma unroll
        for (int t = 0; t < NLOAD; ++t) {
            f2 ab;
            if constexpr (FAST) {
                // conversion and centring by bit pattern (round 5): 0x4B0000bb is the float 2^23 + bb, and ONE packed subtraction of
                // 2^23 + 128 turns both into the centred pixels (ssim_px_sep) -- exact; v_or_b32 x 2 + v_pk_add_f32 instead of
                // v_cvt_f32_ubyte0 x 2 + v_sub_f32 x 2 (9.9 against 14.5 clk at this kernel's three waves per SIMD)
                const f2 biased = {__builtin_bit_cast(float, 0x4B000000u | (uint32_t)ia[t]), __builtin_bit_cast(float, 0x4B000000u | (uint32_t)ib[t])};
                ab = biased - f2{8388736.0f, 8388736.0f};
            } else {
                ab = f2{(float)ia[t], (float)ib[t]};          // retrieve_tile: uint8 -> Float
            }
            const float a = ab.x, b = ab.y;
            const float x = a * b;                            // multiply (exact for 8-bit inputs)
            const int p = sp[t];
            s.ab[p] = ab;
            if constexpr (FAST) {
                s.q[p] = f2{__builtin_fmaf(b, b, a * a), x};      // (a'^2 + b'^2, a'b'): exact integers
            } else {
                s.q[p] = ab * ab;
                xf[2 * p] = x;                          // xx[p].lo
                xf[p > 0 ? 2 * p - 1 : 0] = x;          // xx[p-1].hi (p == 0: rewrites xx[0].lo with the same value)
            }
        }
    };
    auto stage = [&](Slot2& s) { stage_from(s, va, vb); };

    // Accumulator rings (zero == the memset of src/ssim_fma.cpp:187).
    f2 accAB[2][11], accQ[2][11], accX[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        accAB[0][k] = accAB[1][k] = accQ[0][k] = accQ[1][k] = accX[k] = f2{0.0f, 0.0f};
    }
    double colsum[2] = {0.0, 0.0};

    // The first three rows are requested back to back (their memory latencies overlap; one strip of a small
    // image is little more than this prologue plus 18 rows), then staged as they arrive.
    const int r_begin = y0 - 5;
    {
        uint8_t a0[NLOAD], b0[NLOAD], a1[NLOAD], b1[NLOAD];
        fetch_to(r_begin, a0, b0);
        fetch_to(r_begin + 1, a1, b1);
        fetch(r_begin + 2);
        stage_from(ring[0], a0, b0);
        stage_from(ring[1], a1, b1);
    }
    wave_sync();

    // Map addressing, like the loads: uniform row base + non-negative 32-bit byte offset per column.
    const int refM = pd.map_step >= 0 ? x0 : (x0 + Slot2::STRIP_W - 1 < W ? x0 + Slot2::STRIP_W - 1 : W - 1);
    uint32_t offM[2] = {0, 0};
    bool     col_ok[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int x = x0 + 2 * lane + c;
        col_ok[c] = x < W;
        if constexpr (MAP != 0) offM[c] = col_ok[c] ? (uint32_t)((int64_t)(x - refM) * pd.map_step * 4) : 0x80000000u;
    }

    // Window registers.  Index 0 = slot pixel 2 lane + 2 = image column x - 5 of the lane's first column x: column 0 needs
    // window pixels 0..10 (centre 5), column 1 needs 1..11 (centre 6): six 16-byte reads, every half of which is used.
    f2 wab[12], wq[12], wxx[12];
    const int e = 2 * lane + PAD - 5;                   // even: the 16-byte reads are aligned
    auto load_ab = [&](const Slot2& s) {
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const f4 v = *reinterpret_cast<const f4*>(&s.ab[e + 2 * t]);
            wab[2 * t] = v.xy; wab[2 * t + 1] = v.zw;
        }
    };
    // Folded (a,b) sums of the row about to be blurred, and its two centre pixels: computed at the END of the
    // previous iteration (where the window reads they consume are long complete) and carried over the loop edge.
    f2 fa[2][5], ca[2];
    f2 sab[2][6];   // EARLY: the (a,b) streams' row sums S0..S5 of the row about to be scattered
    f2 hab[2];      // separable (a,b) streams (MODE_FAST, MODE_SEPARABLE): the finished ROW pass is what crosses the loop edge (4
                    // registers instead of 24) and its dependent chain of six multiply-adds runs with the fold, in the
                    // low-priority phase of the row: hybrid +2.5...5 % (32 x 4096^2 237.5 -> 243.8 Gpix/s, 32 x 1080p 222 -> 234;
                    // 256 x 1080p -1.3 %), separable +0...1 % and 168 -> 152 VGPRs (profiles/r03_rowpass_ab.txt).  The bit-exact
                    // modes' counterpart is the EARLY form (template parameter): it only pays on short launches.
    auto fold_ab = [&]() {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int m = 5 + c;
            ca[c] = wab[m];
#pragma unroll
            for (int i = 1; i <= 5; ++i) fa[c][i - 1] = wab[m + i] + wab[m - i];   // s[x+i]+s[x-i], src/ssim_fma.cpp:196-201
        }
        if constexpr (HYB || FAST) {
            const f2 s0[6] = {ca[0], fa[0][0], fa[0][1], fa[0][2], fa[0][3], fa[0][4]};
            const f2 s1[6] = {ca[1], fa[1][0], fa[1][1], fa[1][2], fa[1][3], fa[1][4]};
            separable_rows_pair<ORDER_CENTRE_FIRST>(hab[0], hab[1], s0, s1, gf);
        }
        if constexpr (EARLY && !FAST && !HYB) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                sab[c][0] = row_sum<0, FUSED>(ca[c], fa[c][0], fa[c][1], fa[c][2], fa[c][3], fa[c][4]);
                sab[c][1] = row_sum<1, FUSED>(ca[c], fa[c][0], fa[c][1], fa[c][2], fa[c][3], fa[c][4]);
                sab[c][2] = row_sum<2, FUSED>(ca[c], fa[c][0], fa[c][1], fa[c][2], fa[c][3], fa[c][4]);
                sab[c][3] = row_sum<3, FUSED>(ca[c], fa[c][0], fa[c][1], fa[c][2], fa[c][3], fa[c][4]);
                sab[c][4] = row_sum<4, FUSED>(ca[c], fa[c][0], fa[c][1], fa[c][2], fa[c][3], fa[c][4]);
                sab[c][5] = row_sum<5, FUSED>(ca[c], fa[c][0], fa[c][1], fa[c][2], fa[c][3], fa[c][4]);
            }
        }
    };
    load_ab(ring[0]);
    fold_ab();

    // One source row.  `cur` (the LDS slot holding row r) is a compile-time constant: the loops below are unrolled
    // over the two slots by hand, which makes every LDS address an immediate offset and lets the newest ring
    // entry of each stream alternate between two registers instead of being copied into place.
    // `phase`: ROW_WARMUP = one of the 10 rows above the strip's first output row (blur only, nothing finished
    // yet), ROW_MAIN = blur + output row r-5, ROW_LAST = ROW_MAIN without preparing any further row.
    auto row = [&](const int r, auto slot, auto phase_tag, auto kmin_tag) {
        constexpr int cur = decltype(slot)::value;
        constexpr int phase = decltype(phase_tag)::value;
        constexpr int KMIN = decltype(kmin_tag)::value;      // warm-up rows: ring entries below KMIN are not needed (ring_scatter)
        // One wave == one workgroup: wave_sync() only orders LDS accesses for the compiler.
        // LDS latency schedule of one row.  The compiler emits a full s_waitcnt lgkmcnt(0) drain whenever it
        // cannot count (more than 15 operations in flight, or at the loop header), so requests and first uses
        // are placed such that no drain ever waits for much:
        //   - the (a*a,b*b) and ab windows of this row are requested first and land behind the ~100 packed
        //     instructions of the (a,b) streams, which start immediately from the carried-over folds;
        //   - the next row's (a,b) window is requested between the (a*a,b*b) and the ab streams, and is
        //     folded at the bottom of the iteration, BEFORE the staging writes of row r+2 are issued
        //     (a drain at the loop header would otherwise wait for those writes).
        const Slot2& s = ring[cur];
        // The two waves of a SIMD share its VALU; whichever is in its blur phase gets priority over the one
        // that is staging / fetching / folding, so the pair settles into complementary phases instead of
        // contending for the pipe in lockstep (measured +4.6 % exact, +6 % fast and with map, +8 % single image;
        // levels 1-3 are equivalent, dropping before the epilogue instead costs 1.5 %).
        __builtin_amdgcn_s_setprio(2);
        // (1), (2) request the other two planes of this row
        __builtin_amdgcn_sched_barrier(0);
        // Whole 16-byte reads only: 8-byte reads at this 16-byte lane stride are 2-way bank conflicts.  The one end entry
        // the wide form loads without need (the last of the ab plane's pairs) is kept alive until its plane is consumed: a dead
        // half of a wide read gets its register reused and forces an early s_waitcnt.
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const f4 u = *reinterpret_cast<const f4*>(&s.q[e + 2 * t]);
            wq[2 * t] = u.xy;  wq[2 * t + 1] = u.zw;
        }
        if constexpr (!FAST) {
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const f4 z = *reinterpret_cast<const f4*>(&s.xx[e + 2 * t]);
                wxx[2 * t] = z.xy; wxx[2 * t + 1] = z.zw;
            }
        }
        // (3) row sums + ring scatter of the (a,b) streams while those reads are in flight
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!FAST && !HYB && EARLY) {
#pragma unroll
            for (int c = 0; c < 2; ++c) ring_scatter<FUSED, KMIN>(accAB[c], sab[c]);      // on row sums formed at fold time
        } else if constexpr (!FAST && !HYB) {
#pragma unroll
            for (int c = 0; c < 2; ++c) blur_exact<FUSED, KMIN>(accAB[c], ca[c], fa[c][0], fa[c][1], fa[c][2], fa[c][3], fa[c][4]);
        } else {
            separable_columns_pair<KMIN>(accAB[0], accAB[1], hab[0], hab[1], gf);      // the row pass ran with the fold (fold_ab)
        }
        // (4) the (a*a,b*b) streams / MODE_SEPARABLE: the (a'*a' + b'*b', a'b') streams
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!FAST) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int m = 5 + c;
                const f2 q1 = wq[m + 1] + wq[m - 1], q2 = wq[m + 2] + wq[m - 2], q3 = wq[m + 3] + wq[m - 3],
                         q4 = wq[m + 4] + wq[m - 4], q5 = wq[m + 5] + wq[m - 5];
                blur_exact<FUSED, KMIN>(accQ[c], wq[m], q1, q2, q3, q4, q5);
            }
            asm volatile("" :: "v"(wxx[11]));
        } else {
            const f2 s0[6] = {wq[5], wq[6] + wq[4], wq[7] + wq[3], wq[8] + wq[2], wq[9] + wq[1], wq[10] + wq[0]};
            const f2 s1[6] = {wq[6], wq[7] + wq[5], wq[8] + wq[4], wq[9] + wq[3], wq[10] + wq[2], wq[11] + wq[1]};
            blur_separable_pair<ORDER_CENTRE_FIRST, KMIN>(accQ[0], accQ[1], s0, s1, gf);
        }
        Px2 head;
        if constexpr (!FAST) {
            // (5) request the NEXT row's (a,b) window (the other slot was staged an iteration ago).  Not earlier:
            //     with more than 15 LDS operations in flight the compiler can only drain them all.
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (phase != ROW_LAST) load_ab(ring[cur ^ 1]);
            // (6) the ab stream
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (phase != ROW_WARMUP)
                head = ssim_px2_head<!HYB>(accAB[0][0], accAB[1][0], accQ[0][0], accQ[1][0], args.c1, args.c2);
            // ab plane: both columns packed, xx[k] = (ab[k], ab[k+1]); the centre pair is index 5
            const f2 x1 = wxx[6] + wxx[4], x2 = wxx[7] + wxx[3], x3 = wxx[8] + wxx[2], x4 = wxx[9] + wxx[1], x5 = wxx[10] + wxx[0];
            blur_exact<FUSED, KMIN>(accX, wxx[5], x1, x2, x3, x4, x5);
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- ring entry 0 is now the finished output row y = r - 5 (sum_tile) ----
        // No per-row or per-column conditions in here: a conditional epilogue splits the loop body into basic
        // blocks (the compiler then sinks most of the blur below the branch and the scheduling fences lose their
        // meaning), and selects are VALU work.  Rows are sorted into phases by the loops below; columns beyond
        // the image are computed like any other, dropped from the sum at the end and given an out-of-range
        // map offset up front.
        if constexpr (phase != ROW_WARMUP) {
            f2 v;
            if constexpr (FAST) v = ssim_px2_sep(accAB[0][0], accAB[1][0], accQ[0][0], accQ[1][0], args.c1, args.c2);
            else                v = ssim_px2_tail<!HYB>(head, accX[0], args.c1, args.c2);
            colsum[0] += (double)v.x;               // fp64 accumulation, src/ssim_avx.cpp:357-358
            colsum[1] += (double)v.y;
            if constexpr (MAP != 0) {
                // Branch-free store: a raw buffer descriptor over [row base, +2 GiB); lanes with nothing to
                // store present an offset beyond it and the hardware drops the write.
                const int y = r - 5;
                float* mrow = pd.map + ((int64_t)y * pd.map_stride + (int64_t)refM * pd.map_step);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mrow, 0, 0x7FFFFFFF, 0x00020000);
                const float v0 = v.x, v1 = v.y;     // (bit_cast straight from a vector element reads element 0 for both)
                if constexpr (MAP == 2) {
                    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
                    const u2 both = {__builtin_bit_cast(uint32_t, v0), __builtin_bit_cast(uint32_t, v1)};
                    __builtin_amdgcn_raw_buffer_store_b64(both, rs, offM[0], 0, SSIM_MAP_STORE_AUX);     // even width: both columns or neither
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v0), rs, offM[0], 0, SSIM_MAP_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v1), rs, offM[1], 0, SSIM_MAP_STORE_AUX);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (phase != ROW_LAST && !FAST) {
            __builtin_amdgcn_sched_barrier(0);
            fold_ab();                                  // row r+1 (window requested in step (5))
            __builtin_amdgcn_sched_barrier(0);
            wave_sync();
            stage(ring[cur]);                           // row r+2 replaces row r
            fetch(r + 3);
            wave_sync();
        }
        if constexpr (phase != ROW_LAST && FAST) {
            // MODE_SEPARABLE runs three waves per SIMD (168 VGPRs): the next row's (a,b) window is requested only now, so
            // that its 28 registers are not live during the streams above, and folded behind the staging of row r+2
            // (the other two waves cover the latency).
            __builtin_amdgcn_sched_barrier(0);
            load_ab(ring[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            wave_sync();
            stage(ring[cur]);                           // row r+2 replaces row r
            fetch(r + 3);
            wave_sync();
            __builtin_amdgcn_sched_barrier(0);
            fold_ab();                                  // row r+1
        }
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, ROW_WARMUP> Warm;
    typedef std::integral_constant<int, 0> K0;
    int r = r_begin;
    // The ten warm-up rows, in pairs (the two LDS slots): warm-up row i only feeds ring entries k >= 10 - i, so the first
    // three pairs run with KMIN = 9, 7, 5 -- 22 % of the warm-up's arithmetic (a 64-row strip of a lone 4096^2 pair: -3 %
    // of its time; an 8-row strip of a small image: -12 %) -- and the last two as they are (KMIN 3 and 1 would save 4 adds).
    row(r, S0(), Warm(), std::integral_constant<int, 9>());     row(r + 1, S1(), Warm(), std::integral_constant<int, 9>());
    row(r + 2, S0(), Warm(), std::integral_constant<int, 7>()); row(r + 3, S1(), Warm(), std::integral_constant<int, 7>());
    row(r + 4, S0(), Warm(), std::integral_constant<int, 5>()); row(r + 5, S1(), Warm(), std::integral_constant<int, 5>());
    r += 6;
#pragma unroll 1
    for (int i = 0; i < 2; ++i, r += 2) {
        row(r, S0(), Warm(), K0());
        row(r + 1, S1(), Warm(), K0());
    }
    // Main rows, one reduction cell at a time; only the image's last cell can be shorter.
    const int cell_rows = 1 << args.cell_shift;
    uint32_t cell_y = (uint32_t)y0 >> args.cell_shift, parked = 0;
#pragma unroll 1
    for (int left = y_end - y0; left > 0; left -= cell_rows) {
        const int rows = left < cell_rows ? left : cell_rows;
#pragma unroll 1
        for (int i = rows >> 1; i > 0; --i, r += 2) {
            row(r, S0(), std::integral_constant<int, ROW_MAIN>(), K0());
            row(r + 1, S1(), std::integral_constant<int, ROW_MAIN>(), K0());
        }
        if (rows & 1)
            row(r, S0(), std::integral_constant<int, ROW_LAST>(), K0());
        cells.leaf[parked][lane] = (col_ok[0] ? colsum[0] : 0.0) + (col_ok[1] ? colsum[1] : 0.0);      // this cell's leaf of the lane
        colsum[0] = colsum[1] = 0.0;
        if (++parked == CELL_BATCH) {
            wave_sync();
            cell_batch_flush2(args, st, cells, cell_y, parked);
            wave_sync();
            cell_y += parked;
            parked = 0;
        }
    }
    if (parked) {
        wave_sync();
        cell_batch_flush2(args, st, cells, cell_y, parked);
    }
    if constexpr (BAL) wave_sync();     // the next segment re-uses the LDS slots and the cell batch
    } while (BAL && wk.end > wk.first);
    if (SSIM_CLOCKED(args)) clock_end(args.clock);
}

// ---------------------------------------------------------------------------------------------
// ssim_strip1_kernel -- one column per lane, 64-column strips; the same row choreography as the
// two-column kernel (two LDS slots unrolled by hand, warm-up / main / last phases, the next row's
// (a,b) window requested a stream ahead and folded at the bottom of the iteration).  Two jobs:
//  * MODE_DOUBLE: the only shape whose fp64 accumulator rings (110 VGPRs) fit in the register file;
//  * the fully general fallback of the fp32 modes: 64-bit coordinates and per-lane 64-bit offsets
//    (image pairs fits_strip2() rejects) and tuning variant 1.  In fp32 it has half the accumulator
//    registers of the two-column kernel but twice the loader/LDS work per pixel: 10-14 % slower
//    (MODE_SEPARABLE: 25 % slower, although it runs four waves per SIMD there).
// MODE_EXACT / MODE_UNFUSED / MODE_FAST: five planes, the ab plane a scalar stream (MODE_FAST: separable (a,b) stream).
// MODE_SEPARABLE / MODE_DOUBLE: the four planes of ssim_px_sep / ssim_px_four -- (a,b) and (a*a + b*b, ab), centred in
// MODE_SEPARABLE -- as two packed streams; MODE_DOUBLE then needs 160 VGPRs and runs
// three waves per SIMD (five planes: 193 VGPRs, two waves; 8 x 4096^2 163 -> 214 Gpix/s).
// ---------------------------------------------------------------------------------------------
struct Slot1 {
    static constexpr int STRIP_W = 64, PAD = 8, ROW_PX = STRIP_W + 2 * PAD;
    f2    ab[ROW_PX];   // (a, b)
    f2    q[ROW_PX];    // (a*a, b*b)
    float x[ROW_PX];    // a*b
};

// WIDE: 64-bit coordinates and per-lane 64-bit offsets (pairs fits_strip2() rejects).  !WIDE (round 3; every other launch):
// the two-column kernel's addressing -- 32-bit coordinates on the scalar unit, a wave-uniform 64-bit row base plus a
// loop-invariant non-negative 32-bit lane offset per load (global_load "saddr" form), the map through a branch-free raw
// buffer store -- which removes ~10 VALU instructions per pixel of address arithmetic and 64-bit compares from the row
// loop (MODE_DOUBLE: 147 -> 138 issue slots per pixel).
template <int MODE, bool MAP, bool WIDE>
__global__ __launch_bounds__(64) void ssim_strip1_kernel(const KArgs args)
{
    constexpr int PAD = Slot1::PAD, ROW_PX = Slot1::ROW_PX;
    constexpr int NLOAD = 2;
    constexpr bool DBL = (MODE == MODE_DOUBLE);
    constexpr bool FAST = (MODE == MODE_SEPARABLE);
    constexpr bool HYB = (MODE == MODE_FAST);        // exact E[.] streams, separable (a,b) stream
    constexpr bool FOUR = FAST || DBL;               // four planes: (a,b) and (a*a + b*b, ab), see ssim_px_sep / ssim_px_four
    constexpr bool FUSED = (MODE != MODE_UNFUSED);
    typedef typename std::conditional<DBL, d2, f2>::type PV;         // plane-pair streams
    typedef typename std::conditional<DBL, double, float>::type XV;  // ab stream

    __shared__ __attribute__((aligned(16))) Slot1 ring[2];
    __shared__ __attribute__((aligned(16))) CellBatch cells;

    const int lane = threadIdx.x;
    if (SSIM_CLOCKED(args)) clock_begin(args.clock);
    const Strip st = strip_setup(args, Slot1::STRIP_W);
    const PairDesc& pd = st.pd;
    typedef typename std::conditional<WIDE, int64_t, int>::type idx_t;       // coordinates
    typedef typename std::conditional<WIDE, int64_t, uint32_t>::type off_t;  // per-lane byte offsets
    const idx_t W = (idx_t)st.W, H = (idx_t)st.H, x0 = (idx_t)st.x0, y0 = (idx_t)st.y0, y_end = (idx_t)st.y_end;

    // Staging columns: slot pixel p is image column clamp(x0 - PAD + p).  !WIDE: offsets are relative to the strip's
    // lowest-addressed column so that they are non-negative and fit 32 bits (see ssim_strip2_kernel).
    auto clampx = [&](idx_t x) { return x < 0 ? (idx_t)0 : (x > W - 1 ? W - 1 : x); };
    const idx_t x_lo = clampx(x0 - PAD), x_hi = clampx(x0 - PAD + ROW_PX - 1);
    const idx_t refA = WIDE ? (idx_t)0 : (pd.a_step >= 0 ? x_lo : x_hi), refB = WIDE ? (idx_t)0 : (pd.b_step >= 0 ? x_lo : x_hi);
    const gptr_u8 baseA = (gptr_u8)pd.a + (int64_t)refA * pd.a_step;
    const gptr_u8 baseB = (gptr_u8)pd.b + (int64_t)refB * pd.b_step;
    int   sp[NLOAD];
    off_t offA[NLOAD], offB[NLOAD];
#pragma unroll
    for (int t = 0; t < NLOAD; ++t) {
        int p = lane + 64 * t;
        p = p < ROW_PX ? p : ROW_PX - 1;
        const idx_t xg = clampx(x0 - PAD + p);
        sp[t] = p;
        offA[t] = (off_t)((int64_t)(xg - refA) * pd.a_step);
        offB[t] = (off_t)((int64_t)(xg - refB) * pd.b_step);
    }

    uint8_t va[NLOAD], vb[NLOAD];
    auto fetch_to = [&](idx_t r, uint8_t (&oa)[NLOAD], uint8_t (&ob)[NLOAD]) {
        const idx_t ry = r < 0 ? (idx_t)0 : (r > H - 1 ? H - 1 : r);
        const gptr_u8 ra = baseA + (int64_t)ry * pd.a_stride;
        const gptr_u8 rb = baseB + (int64_t)ry * pd.b_stride;
#pragma unroll
        for (int t = 0; t < NLOAD; ++t) {
            if constexpr (!WIDE) asm volatile("" : "+v"(offA[t]), "+v"(offB[t]));   // keeps the zero-extension foldable into the load (ssim_strip2_kernel)
            oa[t] = ra[offA[t]];
            ob[t] = rb[offB[t]];
        }
    };
    auto fetch = [&](idx_t r) { fetch_to(r, va, vb); };
    auto stage_from = [&](Slot1& s, const uint8_t (&ia)[NLOAD], const uint8_t (&ib)[NLOAD]) {
#pragma unroll
        for (int t = 0; t < NLOAD; ++t) {
            float a = (float)ia[t], b = (float)ib[t];
            if constexpr (FAST) { a -= kCentre; b -= kCentre; }   // centred pixels (ssim_px_sep); exact
            const f2 ab = {a, b};
            s.ab[sp[t]] = ab;
            if constexpr (FOUR) {
                s.q[sp[t]] = f2{__builtin_fmaf(b, b, a * a), a * b};
            } else {
                s.q[sp[t]] = ab * ab;
                s.x[sp[t]] = a * b;
            }
        }
    };
    auto stage = [&](Slot1& s) { stage_from(s, va, vb); };

    PV accAB[11], accQ[11];
    XV accX[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        accAB[k] = VT<PV>::splat(0);
        accQ[k] = VT<PV>::splat(0);
        accX[k] = VT<XV>::splat(0);
    }
    double colsum = 0.0;
    const idx_t xcol = x0 + lane;
    const bool col_ok = xcol < W;
    const int64_t map_off = (MAP && WIDE) ? (int64_t)xcol * pd.map_step : 0;
    // !WIDE map addressing, like the loads: uniform row base + non-negative 32-bit byte offset; a lane beyond the image
    // carries an out-of-range offset and the raw buffer store drops it (no exec masking in the row loop)
    const idx_t refM = pd.map_step >= 0 ? x0 : (x0 + Slot1::STRIP_W - 1 < W ? x0 + Slot1::STRIP_W - 1 : W - 1);
    const uint32_t offM = (MAP && !WIDE) ? (col_ok ? (uint32_t)((int64_t)(xcol - refM) * pd.map_step * 4) : 0x80000000u) : 0u;

    const idx_t r_begin = y0 - 5;
    {   // three rows requested back to back, as in the two-column kernel
        uint8_t a0[NLOAD], b0[NLOAD], a1[NLOAD], b1[NLOAD];
        fetch_to(r_begin, a0, b0);
        fetch_to(r_begin + 1, a1, b1);
        fetch(r_begin + 2);
        stage_from(ring[0], a0, b0);
        stage_from(ring[1], a1, b1);
    }
    wave_sync();

    // Window of the lane's column: slot pixels base .. base+10 (centre base+5).
    const int base = lane + PAD - 5;
    f2 wab[11], wq[11];
    float wx[11];
    // The lane's window addresses of both slots, kept in registers: the compiler pairs adjacent 8-byte reads into
    // ds_read2_b64, whose offset field reaches 2040 bytes only -- without the pinned pointers it rebuilds "slot base +
    // constant" with a VALU add for most reads of the second slot, every row (7 v_add_u32 per two rows).
    typedef const f2 __attribute__((address_space(3)))* lptr_f2;
    lptr_f2 pab0 = (lptr_f2)&ring[0].ab[base], pab1 = (lptr_f2)&ring[1].ab[base];
    lptr_f2 pq0 = (lptr_f2)&ring[0].q[base], pq1 = (lptr_f2)&ring[1].q[base];
    asm volatile("" : "+v"(pab0), "+v"(pab1), "+v"(pq0), "+v"(pq1));
    // folded (a,b) sums of the row about to be blurred + its centre pixel, carried over the loop edge
    f2 fa[5], ca;
    auto load_ab = [&](int slot) {
        const lptr_f2 p = slot ? pab1 : pab0;
#pragma unroll
        for (int t = 0; t < 11; ++t) wab[t] = p[t];
    };
    auto fold_ab = [&]() {
        ca = wab[5];
#pragma unroll
        for (int i = 1; i <= 5; ++i) fa[i - 1] = wab[5 + i] + wab[5 - i];   // s[x+i]+s[x-i], src/ssim_fma.cpp:196-201
    };
    load_ab(0);
    fold_ab();

    auto blur = [&](auto& acc, auto s0, auto s1, auto s2, auto s3, auto s4, auto s5, auto mu_stream, auto kmin_tag) {
        constexpr int KMIN = decltype(kmin_tag)::value;       // warm-up rows: see ring_scatter
        if constexpr (MODE == MODE_EXACT || MODE == MODE_UNFUSED || (HYB && !decltype(mu_stream)::value)) blur_exact<FUSED, KMIN>(acc, s0, s1, s2, s3, s4, s5);
        else if constexpr (FAST || HYB)   // the same tap order as the two-column kernel: the two agree bit for bit
            blur_separable<ORDER_CENTRE_FIRST, KMIN>(acc, s0, s1, s2, s3, s4, s5, args.gf);
        else  // fp64 internals: the folded sums are exact integers in fp32; everything after is double
            blur_separable<ORDER_CENTRE_FIRST, KMIN>(acc, to_f64(s0), to_f64(s1), to_f64(s2), to_f64(s3), to_f64(s4), to_f64(s5), args.gd);
    };

    auto row = [&](const idx_t r, auto slot, auto phase_tag, auto kmin_tag) {
        constexpr int cur = decltype(slot)::value;
        constexpr int phase = decltype(phase_tag)::value;
        const Slot1& s = ring[cur];
        __builtin_amdgcn_s_setprio(2);               // see ssim_strip2_kernel
        __builtin_amdgcn_sched_barrier(0);
        const lptr_f2 pq = cur ? pq1 : pq0;
#pragma unroll
        for (int t = 0; t < 11; ++t) {
            wq[t] = pq[t];
            if constexpr (!FOUR) wx[t] = s.x[base + t];
        }
        __builtin_amdgcn_sched_barrier(0);
        blur(accAB, ca, fa[0], fa[1], fa[2], fa[3], fa[4], std::true_type(), kmin_tag);
        __builtin_amdgcn_sched_barrier(0);
        blur(accQ, wq[5], wq[6] + wq[4], wq[7] + wq[3], wq[8] + wq[2], wq[9] + wq[1], wq[10] + wq[0], std::false_type(), kmin_tag);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (phase != ROW_LAST) load_ab(cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!FOUR)
            blur(accX, wx[5], wx[6] + wx[4], wx[7] + wx[3], wx[8] + wx[2], wx[9] + wx[1], wx[10] + wx[0], std::false_type(), kmin_tag);
        __builtin_amdgcn_sched_barrier(0);

        if constexpr (phase != ROW_WARMUP) {         // ring entry 0 is the finished output row y = r - 5
            float vmap;
            if constexpr (DBL) {
                const double v = ssim_px_four(accAB[0].x, accAB[0].y, accQ[0].x, accQ[0].y, args.c1d, args.c2d);
                colsum += v;
                vmap = (float)v;                    // the reference's map is float in the double build too
            } else if constexpr (FAST) {
                const float v = ssim_px_sep(accAB[0].x, accAB[0].y, accQ[0].x, accQ[0].y, args.c1, args.c2);
                colsum += (double)v;
                vmap = v;
            } else {
                const float v = ssim_px<!HYB>(accAB[0].x, accAB[0].y, accQ[0].x, accQ[0].y, accX[0], args.c1, args.c2);
                colsum += (double)v;
                vmap = v;
            }
            if constexpr (MAP && WIDE) {
                if (col_ok) __builtin_nontemporal_store(vmap, &((gptr_f32)pd.map)[(r - 5) * pd.map_stride + map_off]);   // see SSIM_MAP_STORE_AUX
            }
            if constexpr (MAP && !WIDE) {
                float* mrow = pd.map + ((int64_t)(r - 5) * pd.map_stride + (int64_t)refM * pd.map_step);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mrow, 0, 0x7FFFFFFF, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, vmap), rs, offM, 0, SSIM_MAP_STORE_AUX);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (phase != ROW_LAST) {
            __builtin_amdgcn_sched_barrier(0);
            fold_ab();
            __builtin_amdgcn_sched_barrier(0);
            wave_sync();
            stage(ring[cur]);                       // row r+2 replaces row r
            fetch(r + 3);
            wave_sync();
        }
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, ROW_WARMUP> Warm;
    typedef std::integral_constant<int, 0> K0;
    idx_t r = r_begin;
    // warm-up pairs with KMIN = 9, 7, 5, then two unspecialised ones (see ssim_strip2_kernel)
    row(r, S0(), Warm(), std::integral_constant<int, 9>());     row(r + 1, S1(), Warm(), std::integral_constant<int, 9>());
    row(r + 2, S0(), Warm(), std::integral_constant<int, 7>()); row(r + 3, S1(), Warm(), std::integral_constant<int, 7>());
    row(r + 4, S0(), Warm(), std::integral_constant<int, 5>()); row(r + 5, S1(), Warm(), std::integral_constant<int, 5>());
    r += 6;
#pragma unroll 1
    for (int i = 0; i < 2; ++i, r += 2) {
        row(r, S0(), Warm(), K0());
        row(r + 1, S1(), Warm(), K0());
    }
    const int cell_rows = 1 << args.cell_shift;
    uint32_t cell_y = (uint32_t)(y0 >> args.cell_shift), parked = 0;
#pragma unroll 1
    for (idx_t left = y_end - y0; left > 0; left -= cell_rows) {
        const int rows = left < cell_rows ? (int)left : cell_rows;
#pragma unroll 1
        for (int i = rows >> 1; i > 0; --i, r += 2) {
            row(r, S0(), std::integral_constant<int, ROW_MAIN>(), K0());
            row(r + 1, S1(), std::integral_constant<int, ROW_MAIN>(), K0());
        }
        if (rows & 1)
            row(r, S0(), std::integral_constant<int, ROW_LAST>(), K0());
        cells.leaf[parked][lane] = col_ok ? colsum : 0.0;
        colsum = 0.0;
        if (++parked == CELL_BATCH) {
            wave_sync();
            cell_batch_flush1(args, st, cells, cell_y, parked);
            wave_sync();
            cell_y += parked;
            parked = 0;
        }
    }
    if (parked) {
        wave_sync();
        cell_batch_flush1(args, st, cells, cell_y, parked);
    }
    if (SSIM_CLOCKED(args)) clock_end(args.clock);
}

// Per-image sum of the cell partials in a fixed order, so that the result depends on nothing but the image size:
// thread t of 1024 adds partials t, t+1024, ... in that order; each wave then runs a fixed xor butterfly (every lane
// ends with the same bits); the 16 wave totals are added in wave order.  Images with many cells (an 8192^2 pair has
// 32768) are first cut into chunks of kReduceChunk cells, one block each (grid.y), then the chunk sums are summed the
// same way.  One 4096^2 image: 8192 cells = 8 loads per thread (a single 256-thread tree took 8.7 us of a 96 us launch).
constexpr uint32_t kReduceChunk = 8192;
constexpr int      kReduceThreads = 1024;

__global__ __launch_bounds__(kReduceThreads) void ssim_reduce_kernel(const double* __restrict__ partials, uint32_t per_image, uint32_t chunk, double* __restrict__ sums)
{
    __shared__ double sh[kReduceThreads / 64];
    const uint32_t first = blockIdx.y * chunk;
    const uint32_t n = per_image - first < chunk ? per_image - first : chunk;
    const double* p = partials + (size_t)blockIdx.x * per_image + first;
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += kReduceThreads)
        acc += p[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63u) == 0)
        sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sh[0];
#pragma unroll
        for (int w = 1; w < kReduceThreads / 64; ++w)
            t += sh[w];
        sums[(size_t)blockIdx.x * gridDim.y + blockIdx.y] = t;
    }
}

template <int MODE>
hipError_t launch_strip2(const Geometry& geo, const KArgs& ka, bool map, hipStream_t stream, bool early)
{
    const dim3 grid(geo.strips_x, geo.strips_y, geo.count), block(64);
    if constexpr (MODE != MODE_SEPARABLE) {
        if (geo.chunk_cells != 0 && !map) {          // the balanced schedule (plan()): no map; the bit-exact modes with EARLY row sums
            constexpr bool EARLY = (MODE == MODE_EXACT || MODE == MODE_UNFUSED);
            hipLaunchKernelGGL((ssim_strip2_kernel<MODE, 0, EARLY, true>), dim3(geo.n_chunks, 1, 1), block, 0, stream, ka);
            return hipGetLastError();
        }
    }
    if constexpr (MODE == MODE_EXACT || MODE == MODE_UNFUSED) {
        if (early) {
            if (!map)              hipLaunchKernelGGL((ssim_strip2_kernel<MODE, 0, true>), grid, block, 0, stream, ka);
            else if (geo.map_unit) hipLaunchKernelGGL((ssim_strip2_kernel<MODE, 2, true>), grid, block, 0, stream, ka);
            else                   hipLaunchKernelGGL((ssim_strip2_kernel<MODE, 1, true>), grid, block, 0, stream, ka);
            return hipGetLastError();
        }
    }
    if (!map)              hipLaunchKernelGGL((ssim_strip2_kernel<MODE, 0>), grid, block, 0, stream, ka);
    else if (geo.map_unit) hipLaunchKernelGGL((ssim_strip2_kernel<MODE, 2>), grid, block, 0, stream, ka);
    else                   hipLaunchKernelGGL((ssim_strip2_kernel<MODE, 1>), grid, block, 0, stream, ka);
    return hipGetLastError();
}

template <int MODE>
hipError_t launch_strip1(const Geometry& geo, const KArgs& ka, bool map, hipStream_t stream)
{
    const dim3 grid(geo.strips_x, geo.strips_y, geo.count), block(64);
    if (geo.wide) {
        if (map) hipLaunchKernelGGL((ssim_strip1_kernel<MODE, true, true>), grid, block, 0, stream, ka);
        else     hipLaunchKernelGGL((ssim_strip1_kernel<MODE, false, true>), grid, block, 0, stream, ka);
    } else {
        if (map) hipLaunchKernelGGL((ssim_strip1_kernel<MODE, true, false>), grid, block, 0, stream, ka);
        else     hipLaunchKernelGGL((ssim_strip1_kernel<MODE, false, false>), grid, block, 0, stream, ka);
    }
    return hipGetLastError();
}

} // namespace

namespace {

// Y = (R*19595 + G*38470 + B*7471 + 32768) / 65536: the integer BT.601 weights of the reference's
// CLI (src/ssim-cli.cpp:158-186).  A byte kernel, HBM-bound (3-4 B read, 1 B written per pixel): what matters is that a
// wave touches memory in whole dwords.  Pixels of 3 bytes (RGB) or 4 (RGBA / RGBX) are processed FOUR per thread -- three
// or four dwords read, one dword written -- at ANY byte alignment of the rows: global memory takes dword accesses at odd
// addresses on gfx950 (the type below tells the compiler not to assume more), so neither an odd row pitch nor an odd base
// address sends a frame to the per-pixel path (round 4, 8192^2: RGBA 158.8 -> 55.0 us, RGB with an odd row pitch 110.8 -> 39.7 us;
// 16384^2, beyond the 256 MB Infinity Cache: 5.8 / 6.3 / 4.8 TB/s of algorithmic bytes for RGB / RGBA / odd-pitch RGB;
// tools/luminance_probe.py, profiles/r04_luminance_probe.txt).
// Any other step (planar-with-gaps layouts, negative steps) takes one pixel per thread.
__device__ __forceinline__ uint32_t bt601(uint32_t r, uint32_t g, uint32_t b)
{
    return (r * 19595u + g * 38470u + b * 7471u + 32768u) >> 16;
}

typedef uint32_t __attribute__((aligned(1))) u32_any;     // a dword at any byte address

template <int PX>      // bytes per pixel of the four-pixels-per-thread form: 3 or 4; 0 = any step, one pixel per thread
__global__ __launch_bounds__(256) void luminance_kernel(uint8_t* __restrict__ dst, int64_t dst_stride, const uint8_t* __restrict__ src,
                                                       int64_t src_step, int64_t src_stride, uint32_t width, uint32_t height)
{
    const uint32_t y = blockIdx.y;
    const uint8_t* srow = src + (int64_t)y * src_stride;
    uint8_t* drow = dst + (int64_t)y * dst_stride;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (PX == 0) {
        for (uint32_t x = i; x < width; x += gridDim.x * blockDim.x) {
            const uint8_t* px = srow + (int64_t)x * src_step;
            drow[x] = (uint8_t)bt601(px[0], px[1], px[2]);
        }
    } else {
        const uint32_t quads = width >> 2;
        if (i < quads) {
            const u32_any* p = reinterpret_cast<const u32_any*>(srow + (size_t)i * (4 * PX));
            uint32_t y0, y1, y2, y3;
            if constexpr (PX == 3) {
                const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];    // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                y0 = bt601(w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u);
                y1 = bt601(w0 >> 24, w1 & 255u, (w1 >> 8) & 255u);
                y2 = bt601((w1 >> 16) & 255u, w1 >> 24, w2 & 255u);
                y3 = bt601((w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24);
            } else {
                // R G B x, one pixel per dword.  Only bytes 0..2 of a pixel may be read (rmgr_ssim_hip_luminance_device): the dword of a
                // row's LAST pixel would touch byte 3 -- one byte past the caller's buffer when the source pointer is offset into the
                // pixel (ARGB with src = base + 1) and the pixel is the image's last -- so that one pixel is assembled from bytes.
                const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
                uint32_t w3;
                if (4u * i + 4u == width) {
                    const uint8_t* q = reinterpret_cast<const uint8_t*>(p + 3);
                    w3 = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16);
                } else {
                    w3 = p[3];
                }
                y0 = bt601(w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u);
                y1 = bt601(w1 & 255u, (w1 >> 8) & 255u, (w1 >> 16) & 255u);
                y2 = bt601(w2 & 255u, (w2 >> 8) & 255u, (w2 >> 16) & 255u);
                y3 = bt601(w3 & 255u, (w3 >> 8) & 255u, (w3 >> 16) & 255u);
            }
            *reinterpret_cast<u32_any*>(drow + 4 * (size_t)i) = y0 | (y1 << 8) | (y2 << 16) | (y3 << 24);
        } else if (i == quads) {
            for (uint32_t x = quads << 2; x < width; ++x) {
                const uint8_t* px = srow + (int64_t)x * PX;
                drow[x] = (uint8_t)bt601(px[0], px[1], px[2]);
            }
        }
    }
}

} // namespace

hipError_t launch_luminance(uint8_t* dst, int64_t dst_stride, const uint8_t* src, int64_t src_step, int64_t src_stride,
                            uint32_t width, uint32_t height, hipStream_t stream)
{
    if (width == 0 || height == 0) return hipSuccess;
    const bool quad = src_step == 3 || src_step == 4;
    const uint32_t items = quad ? (width >> 2) + 1 : width;
    const dim3 grid((items + 255) / 256, height), block(256);
    if (src_step == 3)      hipLaunchKernelGGL(luminance_kernel<3>, grid, block, 0, stream, dst, dst_stride, src, src_step, src_stride, width, height);
    else if (src_step == 4) hipLaunchKernelGGL(luminance_kernel<4>, grid, block, 0, stream, dst, dst_stride, src, src_step, src_stride, width, height);
    else                    hipLaunchKernelGGL(luminance_kernel<0>, grid, block, 0, stream, dst, dst_stride, src, src_step, src_stride, width, height);
    return hipGetLastError();
}

namespace {

// r = splitmix64(seed ^ ((y << 32) | x)); g = ((3x + 5y) >> 2) & 255; A = (3g + (r & 255)) >> 2;
// B = clamp(A + ((r >> 8) % 33) - 16, 0, 255).  HBM-bound (2 B written per pixel).
__global__ __launch_bounds__(256) void synth_pair_kernel(uint8_t* __restrict__ a, int64_t a_stride, uint8_t* __restrict__ b, int64_t b_stride,
                                                        uint32_t width, uint32_t height, uint64_t seed)
{
    const uint32_t y = blockIdx.y;
    for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < width; x += gridDim.x * blockDim.x) {
        uint64_t z = seed ^ (((uint64_t)y << 32) | x);
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        const uint64_t r = z ^ (z >> 31);
        const uint32_t g = ((3u * x + 5u * y) >> 2) & 255u;
        const int32_t va = (int32_t)((3u * g + (uint32_t)(r & 255u)) >> 2);
        const int32_t n = (int32_t)((r >> 8) % 33u) - 16;
        const int32_t vb = va + n;
        a[(int64_t)y * a_stride + x] = (uint8_t)va;
        b[(int64_t)y * b_stride + x] = (uint8_t)(vb < 0 ? 0 : (vb > 255 ? 255 : vb));
    }
}

} // namespace

hipError_t launch_synth_pair(uint8_t* a, int64_t a_stride, uint8_t* b, int64_t b_stride, uint32_t width, uint32_t height,
                             uint64_t seed, hipStream_t stream)
{
    if (width == 0 || height == 0) return hipSuccess;
    const dim3 grid(std::min<uint32_t>((width + 255) / 256, 64u), height), block(256);
    hipLaunchKernelGGL(synth_pair_kernel, grid, block, 0, stream, a, a_stride, b, b_stride, width, height, seed);
    return hipGetLastError();
}

#ifndef SSIM_KERNELS_SOURCE_ID
#define SSIM_KERNELS_SOURCE_ID "unknown"
#endif
const char* kernels_source_id() { return SSIM_KERNELS_SOURCE_ID; }

// Waves per SIMD each kernel runs at (its VGPR count): what plan() packs strips with.
static int waves_per_simd(int mode, int variant)
{
    (void)variant;
    if (mode == MODE_DOUBLE) return 3;                       // 160 VGPRs
    if (mode == MODE_SEPARABLE) return 3;                    // two columns: 146 VGPRs since round 5 (168 before); one column: 105 (4 waves, planned as 3)
    return 2;                                                // bit-exact modes 224-228 VGPRs, MODE_FAST 208 (round 5; 232 / 227 before); one column without map: 146, planned as 2
}

static int columns_per_lane(int mode, int variant)
{
    if (mode == MODE_DOUBLE) return 1;
    return variant == 1 ? 1 : 2;   // variant 1: one column per lane (lower VGPR use, more waves)
}

Geometry plan(uint32_t width, uint32_t height, uint32_t count, int mode, int strip_rows, int variant, int cu_count, int xcd_count,
              uint32_t y_begin, uint32_t y_rows)
{
    Geometry g;
    g.width = width; g.height = height; g.count = count;
    g.map_unit = false;
    g.wide = true;          // safe default; the caller clears it when every pair passes fits_strip2()
    g.xcds = xcd_count >= 1 ? (uint32_t)xcd_count : 8u;     // the modulus of the kernels' XCD-aware workgroup order (scheduling only)
    g.wave_slots = (uint32_t)((cu_count > 0 ? cu_count : 256) * 4 * waves_per_simd(mode, variant));
    g.strip_w = 64 * columns_per_lane(mode, variant);
    g.strips_x = (width + g.strip_w - 1) / g.strip_w;
    g.cell_rows = cell_rows_for(height);
    const uint32_t cr = g.cell_rows;
    g.cells_x = (width + 63) / 64;
    g.cells_y = (height + cr - 1) / cr;
    // the rows this launch produces: [y_begin, y_begin + y_rows), clipped to the image; y_begin on a cell boundary
    g.y_begin = (y_begin < height ? y_begin : height) & ~(cr - 1);
    g.y_end = (y_rows >= height - g.y_begin) ? height : g.y_begin + y_rows;
    const uint32_t rows_total = g.y_end - g.y_begin;
    auto round_cell = [cr](uint32_t v) { return (v + cr - 1) & ~(cr - 1); };      // up to whole cells
    const bool default_rows = strip_rows <= 0;
    uint64_t strips_cost = 0;          // the chosen strips under the packing model below (row-times x 1000); 0: not modelled
    if (strip_rows <= 0 && rows_total > 0) {
        // Default: the strip height with the lowest cost under a DISCRETE model of how the launch's n strips (of u =
        // rows + 10 halo rows + ~2 rows of setup each) pack onto the chip's wave slots -- fitted to strip-height sweeps
        // (profiles/r02_rows_sweep_*.txt; the continuous model of round 1 was off by up to 36 % for mid-size launches:
        // 16 x 1080p ran 9 strips per column = 2160 waves, 112 more than the 2048 slots, i.e. two rounds):
        //   a SIMD holds `waves` strips at a time (2 for the bit-exact two-column kernel and MODE_FAST, 3 for MODE_SEPARABLE / MODE_DOUBLE);
        //   full rounds of waves x SIMDs strips cost u each; the last, partial round costs u x f(k) where k = how many
        //   waves the fullest SIMD still holds: a wave alone on its SIMD runs 1.46x (2-wave kernels) / 2.05x (3-wave
        //   kernels) faster than in a full house, two of three 1.4x faster.
        // Candidates: every even split of the rows into 1 .. rows/cell strips of whole reduction cells, at most `cap` rows tall (below); taller wins ties.
        const uint64_t cus = (uint64_t)(cu_count > 0 ? cu_count : 256), simds = cus * 4;
        const int waves = waves_per_simd(mode, variant);
        // Round 5, three-wave kernels: a launch that fits ONE round is priced with what the round-5 MODE_SEPARABLE kernel measures for strips
        // alone / two / three on a SIMD (1024 / 2048 / 3072 strips of 512 rows: 0.450 / 0.665 / 1 -- two waves per SIMD deliver the throughput
        // of three, so fewer, taller strips win: 8 x 4096^2 and 2 x 8192^2 +5 %, 32 x 1080p +4 %, 256 x 512^2 +2 %); launches of several rounds
        // keep the factors they were fitted with (the new ones lose 1...2 % there), and so does MODE_DOUBLE (-1 % with them): profiles/r05_tail3_sweep.txt.
        static const uint32_t tail2[2] = {685, 1000}, tail3[3] = {490, 715, 1000}, tail3_one_round[3] = {450, 665, 1000};      // x 1/1000
        const uint64_t slots = simds * (uint64_t)(waves >= 3 ? 3 : 2);
        uint64_t best = ~(uint64_t)0;
        // Round 5: strips of the two-wave kernels may be up to 1024 rows tall (512 since round 2: "taller measured no better" then).  Measured on the round-5
        // kernels, 512 / 1024 / 2048 interleaved over 36 shapes (profiles/r05_strip_cap_sweep.txt): 1024 is +1.2 % on 32 and 48 x 4096^2, +1.8 % on 8 x 8192^2
        // (+2.1 % with the map), +0...1.8 % for MODE_FAST, nothing below -1.1 %; 2048 adds a few tenths and loses 2 % on 128 x 4096^2.  The three-wave
        // kernels keep 512 (MODE_SEPARABLE at 1024 / 2048 rows: -3...-4 %).  (Round 5's balanced rule compared with strips of at most 512 rows; round 6's
        // compares with the split chosen here.)
        const uint32_t cap = waves >= 3 ? 512u : 1024u;
        uint32_t best_rows = round_cell(rows_total < cap ? rows_total : cap);
        const uint32_t ny_min = (rows_total + cap - 1) / cap, ny_max = (rows_total + cr - 1) / cr;
        for (uint32_t ny = ny_min; ny <= ny_max; ++ny) {
            const uint32_t rows = round_cell((rows_total + ny - 1) / ny);
            const uint32_t ny_eff = (rows_total + rows - 1) / rows;
            if (ny_eff != ny) continue;                                   // the same split as a smaller ny
            const uint64_t n = (uint64_t)g.strips_x * ny * count, u = rows + 12;
            const uint64_t full = n / slots, rem = n % slots;
            const uint32_t* tail = waves >= 3 ? (full == 0 && mode == MODE_SEPARABLE ? tail3_one_round : tail3) : tail2;
            const uint64_t last = rem == 0 ? 0 : tail[(rem - 1) / simds];
            const uint64_t cost = u * (full * 1000 + last);
            if (cost < best) { best = cost; best_rows = rows; }
        }
        strip_rows = (int)best_rows;
        strips_cost = best;
    }
    if (strip_rows < 1) strip_rows = 1;
    g.strip_rows = round_cell((uint32_t)strip_rows);      // strips start on cell boundaries
    g.strips_y = rows_total ? (rows_total + g.strip_rows - 1) / g.strip_rows : 0;
    // The balanced schedule (work_setup()): the two-waves-per-SIMD two-column kernels (bit-exact modes, MODE_FAST) without a map (launch_strip2() checks the map).  Tuning
    // variant 6 forces it.  By default it is taken where the packing model prices the best strips at or above one round of equal chunks (a chunk: its rows + 12 row-times
    // per segment, 1 + chunk / column of them) -- 4 % above where no interleave of the images brings neighbouring strip columns within 16 rows of each other -- and
    // where a chunk divides the strip column evenly.
    // History of the rule.  Round 5 fitted it on kernels whose chunk list fetched 2.9...3.0x the algorithmic bytes (neighbouring columns out of step: work_setup()):
    // the chunks then ran ~5.5 % behind the model, lost beyond 1100 rows and beyond one column per chunk, and the rule asked for 3.5 % and both caps
    // (profiles/r05_balanced_sweep.txt, r05_rule_sweep.txt).  With the images interleaved (round 6) the chunks run ~2.5 % behind the model and win on long launches as
    // well: rmgr_ssim_hip_tune over 148 launch shapes, 480p ... 8K, 1 ... 256 pairs (tools/tune_sweep.py, profiles/r06_tune_sweep.txt) found the round-5 rule within 1 % of the
    // best candidate on 73 % of them (mean regret 1.0105; 192 / 256 x 1080p +5.4 / +3.2 %, 12 x 8192^2 +7.6 %, 64 x 2160p +8.8 %, 32 x 5K +11.7 % left on the table,
    // all on multi-round strips against one round of chunks).  Re-fitted on that sweep: threshold 0, no caps -- mean regret of the strips-or-chunks choice 1.0072 -> 1.0018.
    // What the chunks still lose on and the rule leaves alone: launches of a few strips per SIMD (2 x 1080p -17 %, 4 x 1000^2 -13 %: ten warm-up rows per 16...64-row chunk).
    // MODE_FAST takes the same rule; MODE_SEPARABLE (three waves per SIMD) loses with chunks nearly everywhere (-1...-17 %; at best +2 %) and has no balanced instantiation.
    g.chunk_cells = 0; g.n_chunks = 0; g.bal_stride = 1;
    if ((mode == MODE_EXACT || mode == MODE_UNFUSED || mode == MODE_FAST) && g.strip_w == 128 && rows_total > 0 && (is_balanced_variant(variant) || (variant == 0 && default_rows))) {
        const uint64_t col_cells = (rows_total + cr - 1) / cr, all = (uint64_t)count * g.strips_x * col_cells;
        const uint64_t want = g.wave_slots;
        if (all > want && all < (1ull << 31)) {
            const uint64_t chunk = (all + want - 1) / want, n_chunks = (all + chunk - 1) / chunk;
            // The phase of the chunks (work_setup()): neighbouring strip columns of an image are C mod c cell rows out of step in the plain list.
            // T images are interleaved column by column, T (C mod c) as close to a multiple of c as an interleave of up to a quarter of what one
            // XCD walks at a time allows (neighbours, T list positions apart, must mostly stay on one XCD); the smallest such T.  Tuning variant
            // 7 forces the plain list (round 5), 100 + T any interleave (measurement aids: tools/phase_ab.sh).
            const uint64_t d = col_cells % chunk;
            uint32_t best_t = 1;
            uint64_t best_dist = 0;
            if (d != 0) {
                const uint64_t per_xcd = (n_chunks / g.xcds) * chunk / col_cells;      // list positions (columns) one XCD holds at a time
                const uint64_t t_max = std::min<uint64_t>(std::min<uint64_t>(count, 64), std::max<uint64_t>(per_xcd / 4, 1));
                best_dist = std::min(d, chunk - d);
                for (uint64_t t = 2; t <= t_max; ++t) {
                    const uint64_t r = (t * d) % chunk, dist = std::min(r, chunk - r);
                    if (dist < best_dist) { best_dist = dist; best_t = (uint32_t)t; }
                }
            }
            bool take = is_balanced_variant(variant);
            if (!take && strips_cost != 0) {
                // one round of n_chunks <= wave slots chunks of chunk x cell rows, + 12 row-times per segment (1 + chunk / col_cells of them)
                const uint64_t simds = (uint64_t)(cu_count > 0 ? cu_count : 256) * 4;
                const uint64_t tail = n_chunks > simds ? 1000 : 685;
                const uint64_t chunks_cost = (chunk * cr * col_cells + 12 * (col_cells + chunk)) * tail / col_cells;
                take = strips_cost * 1000 >= chunks_cost * (best_dist * cr > 16 ? 1040 : 1000);
            }
            // ... and where a chunk divides the strip column evenly: no chunk straddles two columns, so the chunks ARE strips -- the tallest that fill the
            // wave slots in ONE round, without a seam between rounds (8 ... 64 x 4096^2 +0.5 / +0.7 / +1.1 / +2.3 %, 2 ... 16 x 8192^2 +0.9 ... +2.1 %, 32 / 128 x
            // 2048^2 +0.8 / +1.2 %; against the shipped strips on 38 shapes: +0.6 ... +2.4 %, single pairs included; profiles/r05_strip_cap_sweep.txt, last section).
            if (!take && strips_cost != 0 && mode != MODE_FAST && col_cells % chunk == 0) take = true;      // MODE_FAST: -0.8 ... +1.0 %, no net gain: left on its strips
            if (take) {
                g.chunk_cells = (uint32_t)chunk;
                g.n_chunks = (uint32_t)n_chunks;
                g.bal_stride = variant == 7 ? 1u : variant >= 100 ? std::min<uint32_t>((uint32_t)variant - 100u, count) : best_t;
                if (g.bal_stride < 1) g.bal_stride = 1;
            }
        }
    }
    return g;
}

size_t partials_size(const Geometry& geo)
{
    const size_t per = geo.partials_per_image();
    const size_t chunks = per > kReduceChunk ? (per + kReduceChunk - 1) / kReduceChunk : 0;
    return (size_t)geo.count * (per + chunks) + 1;
}

size_t reduce_scratch_size(const Geometry& geo)
{
    const size_t per = geo.partials_per_image();
    return per > kReduceChunk ? (size_t)geo.count * ((per + kReduceChunk - 1) / kReduceChunk) : 0;
}

hipError_t launch_reduce(const Geometry& geo, const double* partials, double* chunk_sums, double* sums, hipStream_t stream)
{
    if (geo.count == 0) return hipSuccess;
    const uint32_t per = geo.partials_per_image();
    if (per == 0) return hipMemsetAsync(sums, 0, sizeof(double) * geo.count, stream);
    if (per > kReduceChunk) {
        const uint32_t chunks = (per + kReduceChunk - 1) / kReduceChunk;
        hipLaunchKernelGGL(ssim_reduce_kernel, dim3(geo.count, chunks), dim3(kReduceThreads), 0, stream, partials, per, kReduceChunk, chunk_sums);
        hipLaunchKernelGGL(ssim_reduce_kernel, dim3(geo.count, 1), dim3(kReduceThreads), 0, stream, chunk_sums, chunks, chunks, sums);
    } else {
        hipLaunchKernelGGL(ssim_reduce_kernel, dim3(geo.count, 1), dim3(kReduceThreads), 0, stream, partials, per, per, sums);
    }
    return hipGetLastError();
}

hipError_t launch(const Geometry& geo, int mode, int variant, int group, const PairDesc* descs_dev, const PairDesc& single,
                  double* partials, double* sums, hipStream_t stream, hipEvent_t ev_begin, hipEvent_t ev_end, bool reduce, uint64_t* clock)
{
    if (geo.count == 0) return hipSuccess;
    KArgs ka;
    ka.single = single;
    ka.descs = descs_dev;
    ka.width = geo.width; ka.height = geo.height;
    ka.strip_rows = geo.strip_rows; ka.strips_x = geo.strips_x; ka.strips_y = geo.strips_y;
    ka.y_begin = geo.y_begin; ka.y_end = geo.y_end;
    ka.cells_x = geo.cells_x; ka.cells_y = geo.cells_y;
    ka.cell_shift = geo.cell_rows == 32 ? 5 : geo.cell_rows == 16 ? 4 : 3;
    ka.col_cells = geo.y_end > geo.y_begin ? (geo.y_end - geo.y_begin + geo.cell_rows - 1) / geo.cell_rows : 0;
    ka.chunk_cells = geo.chunk_cells;
    ka.n_chunks = geo.n_chunks;
    ka.bal_stride = geo.bal_stride >= 1 ? geo.bal_stride : 1u;
    ka.xcds = geo.xcds >= 1 ? geo.xcds : 8u;
    ka.count = geo.count;
    ka.group = (group > 1 && geo.count % (uint32_t)group == 0) ? (uint32_t)group : 1u;
    ka.partials = partials;
    ka.clock = clock;
    // c1, c2: products in double, then cast (src/ssim.cpp:956-960)
    ka.c1d = (0.01 * 255.0) * (0.01 * 255.0);
    ka.c2d = (0.03 * 255.0) * (0.03 * 255.0);
    ka.c1 = (float)ka.c1d;
    ka.c2 = (float)ka.c2d;
    // True 1-D Gaussian, sigma 1.5, normalised over the 11 taps.
    {
        double g[6], norm = 0.0;
        for (int i = 0; i <= 5; ++i) {
            g[i] = exp(-(double)(i * i) / (2.0 * 1.5 * 1.5));
            norm += (i == 0) ? g[i] : 2.0 * g[i];
        }
        for (int i = 0; i <= 5; ++i) {
            ka.gd[i] = g[i] / norm;
            ka.gf[i] = (float)ka.gd[i];
        }
    }
    bool map = single.map != nullptr;   // for batches the ABI guarantees all-or-none and mirrors it into `single`
    if (geo.strips_x == 0 || geo.strips_y == 0)
        return reduce ? launch_reduce(geo, partials, partials + (size_t)geo.count * geo.partials_per_image(), sums, stream) : hipSuccess;
    if (ev_begin) { hipError_t e = hipEventRecord(ev_begin, stream); if (e != hipSuccess) return e; }
    hipError_t err;
    // variant 0: two columns per lane (ssim_strip2_kernel); 1: one column per lane (ssim_strip1_kernel); tuning: 2 forces the
    // two-column kernel with its row sums in the blur phase, 3 with EARLY row sums.  MODE_DOUBLE always runs one column per lane.
    const bool one = columns_per_lane(mode, variant) == 1;
    const bool early = uses_early_row_sums(geo, mode, variant);
    switch (mode) {
    case MODE_EXACT:   err = one ? launch_strip1<MODE_EXACT>(geo, ka, map, stream)   : launch_strip2<MODE_EXACT>(geo, ka, map, stream, early);   break;
    case MODE_UNFUSED: err = one ? launch_strip1<MODE_UNFUSED>(geo, ka, map, stream) : launch_strip2<MODE_UNFUSED>(geo, ka, map, stream, early); break;
    case MODE_FAST:    err = one ? launch_strip1<MODE_FAST>(geo, ka, map, stream)    : launch_strip2<MODE_FAST>(geo, ka, map, stream, false);    break;
    case MODE_SEPARABLE: err = one ? launch_strip1<MODE_SEPARABLE>(geo, ka, map, stream) : launch_strip2<MODE_SEPARABLE>(geo, ka, map, stream, false); break;
    case MODE_DOUBLE:  err = launch_strip1<MODE_DOUBLE>(geo, ka, map, stream); break;
    default:           return hipErrorInvalidValue;
    }
    if (err != hipSuccess) return err;
    if (ev_end) { hipError_t e = hipEventRecord(ev_end, stream); if (e != hipSuccess) return e; }
    return reduce ? launch_reduce(geo, partials, partials + (size_t)geo.count * geo.partials_per_image(), sums, stream) : hipSuccess;
}

} // namespace ssim_hip
