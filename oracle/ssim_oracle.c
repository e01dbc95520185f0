/*
 * ssim_oracle.c -- CPU restatement of rmgr::ssim::compute_ssim() (romigrou/ssim 2.1.0).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it.
 * Nothing under ssim_amd/ (the product) links or falls back to it.
 *
 * Parity pin: validated in the build container against
 *   (1) the real reference FMA/AVX kernel TUs + tests/ssim_naive.h compiled from
 *       /root/reference into oracle/_ref (see oracle/Makefile, oracle/ref_harness.cpp),
 *       bit-for-bit on every committed fixture (tests/test_oracle_vs_ref.py), and
 *   (2) the golden values the reference's own tests hold
 *       (tests/rmgr-ssim-tests.cpp:352-360, einstein set) and the reference outputs recorded
 *       in SURVEY.md A.5 / 8(d) (tests/test_oracle_golden.py).
 *
 * Every function cites the reference file:line it follows.  Nothing here is copied: the
 * reference processes 256x64 tiles through six scratch buffers and SIMD intrinsics; this
 * restatement walks 256x64 tiles with a per-row scatter into plain row buffers, using the same
 * floating-point operations in the same order, which is what bit-exactness needs.
 *
 * Build: gcc -O3 -std=c99 -ffp-contract=off -fopenmp -mfma -mavx2 -shared -fPIC   (oracle/Makefile)
 *   -ffp-contract=off is REQUIRED: fusion happens only where fmaf() is written.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RADIUS 5
#define BAND_H 64   /* = TILE_MAX_HEIGHT, src/ssim.cpp:231 */
#define TILE_W 256  /* = TILE_MAX_WIDTH (float build), src/ssim.cpp:230 */

/*
 * The 21 unique taps of the 11x11 sigma=1.5 window as the SIMD paths hold them
 * (src/ssim_fma.cpp:169-174 == src/ssim_avx.cpp:184-192): triangular order K(i,j), i<=j,
 * index j(j+1)/2+i.  They are the float-computed kernel of src/ssim.cpp:272-318 printed with
 * 18 digits; oracle_generic_kernel21() below recomputes them the generic way and
 * tests/test_oracle_golden.py checks both agree bit-for-bit on glibc.
 */
static const double K21_LITERAL[21] = {
    7.07622393965721130e-02,
    5.66619709134101868e-02, 4.53713610768318176e-02,
    2.90912277996540070e-02, 2.32944320887327194e-02, 1.19597595185041428e-02,
    9.57662798464298248e-03, 7.66836293041706085e-03, 3.93706932663917542e-03, 1.29605561960488558e-03,
    2.02135881409049034e-03, 1.61857774946838617e-03, 8.31005279906094074e-04, 2.73561221547424793e-04, 5.77411265112459660e-05,
    2.73561221547424793e-04, 2.19050692976452410e-04, 1.12464345875196159e-04, 3.70224843209143728e-05, 7.81441485742107034e-06, 1.05756600987660931e-06
};

static inline int tri(int i, int j) { return (i <= j) ? j * (j + 1) / 2 + i : i * (i + 1) / 2 + j; }

/* Literal taps as floats (the `f` suffix of VCOEFF, src/ssim_fma.cpp:92). */
void oracle_kernel21_f32(float out[21])
{
    for (int n = 0; n < 21; ++n)
        out[n] = (float)K21_LITERAL[n];
}

/*
 * Generic-path kernel, float build: src/ssim.cpp:272-318.  Each tap is
 * expf(-(x^2+y^2)/(2 s^2)) / (float(2 pi) s^2), the 121 taps are summed in double in
 * raster order (mirrored taps re-added, :291-300) and every tap divided by float(sum).
 */
void oracle_generic_kernel21_f32(float out[21])
{
    float k[11][11];
    const float sigma = 1.5f;
    const float sigma2 = sigma * sigma;
    double sum = 0.0;
    for (int y = 0; y <= RADIUS; ++y) {
        for (int x = 0; x <= RADIUS; ++x) {
            const int dx = x - RADIUS, dy = y - RADIUS;
            const float num = expf(-(float)(dx * dx + dy * dy) / (2 * sigma2));
            const float den = (float)(2 * M_PI) * sigma2;
            sum += k[y][x] = num / den;
        }
        for (int x = RADIUS + 1; x < 11; ++x)
            sum += k[y][x] = k[y][10 - x];
    }
    for (int y = RADIUS + 1; y < 11; ++y)
        for (int x = 0; x < 11; ++x)
            sum += k[y][x] = k[10 - y][x];
    for (int j = 0; j <= RADIUS; ++j)
        for (int i = 0; i <= j; ++i)
            out[tri(i, j)] = k[RADIUS + j][RADIUS + i] / (float)sum;
}

/*
 * Double kernel of the naive oracle / generic-double path: tests/ssim_naive.h:68-114
 * (== src/ssim.cpp:272-318 with Float=double).  Full 11x11, row-major.
 */
void oracle_kernel121_f64(double out[121])
{
    const double sigma = 1.5, sigma2 = sigma * sigma;
    const double tau = 2 * M_PI;
    double sum = 0.0;
#define KK(x, y) out[(y) * 11 + (x)]
    for (int y = 0; y <= RADIUS; ++y) {
        for (int x = 0; x <= RADIUS; ++x) {
            const int dx = x - RADIUS, dy = y - RADIUS;
            sum += KK(x, y) = exp(-(double)(dx * dx + dy * dy) / (2 * sigma2)) / (tau * sigma2);
        }
        for (int x = RADIUS + 1; x < 11; ++x)
            sum += KK(x, y) = KK(10 - x, y);
    }
    for (int y = RADIUS + 1; y < 11; ++y)
        for (int x = 0; x < 11; ++x)
            sum += KK(x, y) = KK(x, 10 - y);
#undef KK
    for (int n = 0; n < 121; ++n)
        out[n] /= sum;
}

static inline int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------------------------
 * Synthetic input generator, SURVEY.md 8(d): integer-only, bit-reproducible.
 * ------------------------------------------------------------------------------------------ */
static inline uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void oracle_synth_pair(uint8_t* A, uint8_t* B, uint32_t width, uint32_t height, uint64_t seed)
{
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < (int64_t)height; ++y) {
        for (uint32_t x = 0; x < width; ++x) {
            const uint64_t r = splitmix64(seed ^ (((uint64_t)y << 32) | (uint64_t)x));
            const int g = (int)(((3u * x + 5u * (uint32_t)y) >> 2) & 255u);
            const int a = (3 * g + (int)(r & 255)) >> 2;
            const int n = (int)((r >> 8) % 33) - 16;
            const int b = a + n;
            A[(size_t)y * width + x] = (uint8_t)a;
            B[(size_t)y * width + x] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * fp32 path.  fused=1: FMA implementation (blur src/ssim_fma.cpp:196-257, products
 * src/ssim_avx.cpp:76-106, SSIM + fp64 sum src/ssim_avx.cpp:299-410).  fused=0: the AVX /
 * SSE / generic arithmetic (blur with separately rounded mul and add, src/ssim.cpp:404-462).
 * ------------------------------------------------------------------------------------------ */

typedef struct {
    uint32_t width, height;
    const uint8_t* a; ptrdiff_t a_step, a_stride;
    const uint8_t* b; ptrdiff_t b_step, b_stride;
    float* map; ptrdiff_t map_step, map_stride;
    float k[21];
    float c1, c2;
    int fused;
} Job32;

/* One source row r of the 5 statistic planes, edge-replicated by 5 px either side:
 * retrieve_tile (src/ssim.cpp:515-583: halo = nearest edge pixel of the IMAGE) and
 * multiply (src/ssim.cpp:249-265; products are exact for 8-bit inputs). */
static void load_row32(const Job32* jb, int64_t r, int64_t x0, int64_t tw, float* pa, float* pb, float* paa, float* pbb, float* pab)
{
    const int64_t W = jb->width;
    const int64_t ry = clampi(r, 0, (int64_t)jb->height - 1);
    const uint8_t* ra = jb->a + ry * jb->a_stride;
    const uint8_t* rb = jb->b + ry * jb->b_stride;
    for (int64_t xe = 0; xe < tw + 2 * RADIUS; ++xe) {
        const int64_t x = clampi(x0 + xe - RADIUS, 0, W - 1);
        const float a = (float)ra[x * jb->a_step];
        const float b = (float)rb[x * jb->b_step];
        pa[xe] = a;
        pb[xe] = b;
        paa[xe] = a * a;
        pbb[xe] = b * b;
        pab[xe] = a * b;
    }
}

/* The six per-row partial sums S_j(x), j=0..5, of one plane.  Horizontal fold
 * s[x+i]+s[x-i] (src/ssim_fma.cpp:196-201), then sum_j = s0*K(0,j) followed by five
 * multiply-adds in i order (:203-243).  fused: one rounding per step (VFMADD);
 * unfused: MUL_ADD(a,b,c) = a + b*c with both roundings (src/ssim.cpp:353).
 * Written as plain loops over x on precomputed fold rows so that gcc vectorises them (the
 * arithmetic per element is unchanged; -ffp-contract=off keeps the unfused form unfused). */
static void row_sums32(const float* restrict p, int64_t W, const float* restrict k, int fused,
                       float* restrict S /* [6][W] */, float* restrict F /* scratch [5][W] */)
{
    const float* restrict c = p + RADIUS;
    for (int i = 1; i <= RADIUS; ++i) {
        float* restrict f = F + (size_t)(i - 1) * W;
        for (int64_t x = 0; x < W; ++x)
            f[x] = c[x + i] + c[x - i];
    }
    const float* restrict f1 = F, *restrict f2 = F + W, *restrict f3 = F + 2 * W, *restrict f4 = F + 3 * W, *restrict f5 = F + 4 * W;
    for (int j = 0; j <= RADIUS; ++j) {
        const float k0 = k[tri(0, j)], k1 = k[tri(1, j)], k2 = k[tri(2, j)], k3 = k[tri(3, j)], k4 = k[tri(4, j)], k5 = k[tri(5, j)];
        float* restrict s = S + (size_t)j * W;
        if (fused) {
            for (int64_t x = 0; x < W; ++x) {
                float acc = c[x] * k0;
                acc = fmaf(f1[x], k1, acc);
                acc = fmaf(f2[x], k2, acc);
                acc = fmaf(f3[x], k3, acc);
                acc = fmaf(f4[x], k4, acc);
                acc = fmaf(f5[x], k5, acc);
                s[x] = acc;
            }
        } else {
            for (int64_t x = 0; x < W; ++x) {
                float acc = c[x] * k0;
                acc = acc + f1[x] * k1;
                acc = acc + f2[x] * k2;
                acc = acc + f3[x] * k3;
                acc = acc + f4[x] * k4;
                acc = acc + f5[x] * k5;
                s[x] = acc;
            }
        }
    }
}

/* Tile sum in the exact order of the AVX sum_tile (src/ssim_avx.cpp:329-404): per row,
 * groups of 8 pixels go to eight double lanes (Lo = px 0-3, Hi = px 4-7), lanes are folded
 * (Lo+Hi, then 128-bit halves, then two scalar adds into tileSum), remainder pixels are
 * added to tileSum one by one. */
static double tile_sum_avx_order(const float* ssim, int64_t stride, int64_t tw, int64_t th)
{
    double tileSum = 0.0;
    for (int64_t y = 0; y < th; ++y) {
        const float* row = ssim + y * stride;
        int64_t x = 0;
        if (tw >= 8) {
            double lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
            for (; x + 8 <= tw; x += 8) {
                for (int l = 0; l < 4; ++l) {
                    lo[l] += (double)row[x + l];
                    hi[l] += (double)row[x + 4 + l];
                }
            }
            double s[4];
            for (int l = 0; l < 4; ++l)
                s[l] = lo[l] + hi[l];
            const double r0 = s[0] + s[2], r1 = s[1] + s[3];
            tileSum += r0;
            tileSum += r1;
        }
        for (; x < tw; ++x)
            tileSum += (double)row[x];
    }
    return tileSum;
}

/* One tile [x0, x0+tw) x [y0, y0+bh) (at most 256 x 64, the reference's tile so that the working set
 * stays in cache): blur the five planes by row scatter, then the per-pixel SSIM.  Writes the tile's
 * SSIM values to `ssim` (bh x tw, dense) and to the caller's map.  Results do not depend on the tiling:
 * the halo is the image's, not the tile's. */
static void tile32(const Job32* jb, int64_t x0, int64_t tw, int64_t y0, int64_t bh, float* work, float* ssim)
{
    const int64_t EW = tw + 2 * RADIUS;
    float* rows = work;               /* 5 planes x EW    */
    float* S = rows + 5 * EW;         /* 6 x tw           */
    float* F = S + 6 * tw;            /* 5 x tw fold rows */
    float* dst = F + 5 * tw;          /* 5 planes x bh x tw */
    memset(dst, 0, sizeof(float) * 5 * (size_t)bh * (size_t)tw); /* memset per dest row, src/ssim_fma.cpp:187 */

    for (int64_t r = y0 - RADIUS; r < y0 + bh + RADIUS; ++r) {
        load_row32(jb, r, x0, tw, rows, rows + EW, rows + 2 * EW, rows + 3 * EW, rows + 4 * EW);
        for (int p = 0; p < 5; ++p) {
            row_sums32(rows + p * EW, tw, jb->k, jb->fused, S, F);
            float* dp = dst + (size_t)p * bh * tw;
            /* scatter: rows r-5..r+5 of dest += sum5,sum4,..,sum0,..,sum5 (src/ssim_fma.cpp:246-257).
             * Each dest row therefore receives its 11 addends in source-row order, top first. */
            for (int dy = -RADIUS; dy <= RADIUS; ++dy) {
                const int64_t y = r - dy; /* source row r is dest row y's (r-y)-th neighbour */
                if (y < y0 || y >= y0 + bh)
                    continue;
                const int j = dy < 0 ? -dy : dy;
                float* restrict d = dp + (y - y0) * tw;
                const float* restrict sj = S + j * tw;
                for (int64_t x = 0; x < tw; ++x)
                    d[x] = sj[x] + d[x];
            }
        }
    }

    const float c1 = jb->c1, c2 = jb->c2;
    const float* muA = dst;
    const float* muB = dst + (size_t)bh * tw;
    const float* eAA = dst + 2 * (size_t)bh * tw;
    const float* eBB = dst + 3 * (size_t)bh * tw;
    const float* eAB = dst + 4 * (size_t)bh * tw;
    for (int64_t y = 0; y < bh; ++y) {
        for (int64_t x = 0; x < tw; ++x) {
            /* src/ssim.cpp:681-693 == src/ssim_avx.cpp:342-352 (the AVX form negates both
             * sigma factors, which cancels exactly in the quotient). */
            const size_t o = (size_t)y * tw + x;
            const float a = muA[o], b = muB[o];
            const float muA2 = a * a, muB2 = b * b, muAB = a * b;
            const float sA2 = eAA[o] - muA2, sB2 = eBB[o] - muB2, sAB = eAB[o] - muAB;
            const float num = (2 * muAB + c1) * (2 * sAB + c2);
            const float den = (muA2 + muB2 + c1) * (sA2 + sB2 + c2);
            const float v = num / den;
            ssim[o] = v;
            if (jb->map)
                jb->map[(y0 + y) * jb->map_stride + (x0 + x) * jb->map_step] = v;
        }
    }
}

/*
 * Whole-image fp32 SSIM.  Returns the errno-style code of compute_ssim (src/ssim.cpp:962-978).
 * sum_out receives the fp64 sum before the division, so tests can compare partial sums.
 * threads<=1: serial tile order (src/ssim.cpp:1084-1086).
 */
int oracle_ssim_f32(float* ssim_out, double* sum_out, uint32_t width, uint32_t height,
                    const uint8_t* a, ptrdiff_t a_step, ptrdiff_t a_stride,
                    const uint8_t* b, ptrdiff_t b_step, ptrdiff_t b_stride,
                    float* map, ptrdiff_t map_step, ptrdiff_t map_stride, int fused, int threads)
{
    if (!a || !b)
        return 22; /* EINVAL */
    Job32 jb;
    jb.width = width; jb.height = height;
    jb.a = a; jb.a_step = a_step; jb.a_stride = a_stride;
    jb.b = b; jb.b_step = b_step; jb.b_stride = b_stride;
    jb.map = map; jb.map_step = map ? map_step : 0; jb.map_stride = map ? map_stride : 0;
    oracle_kernel21_f32(jb.k);
    jb.c1 = (float)((0.01 * 255.0) * (0.01 * 255.0)); /* src/ssim.cpp:956-960 */
    jb.c2 = (float)((0.03 * 255.0) * (0.03 * 255.0));
    jb.fused = fused;

    const int64_t W = width, H = height;
    const int64_t bands = (H + BAND_H - 1) / BAND_H;
    const int64_t tilesX = (W + TILE_W - 1) / TILE_W;
    double* tileSums = (double*)calloc((size_t)(bands * tilesX > 0 ? bands * tilesX : 1), sizeof(double));
    if (!tileSums)
        return 12;
    int err = 0;
    if (threads < 1)
        threads = 1;
#pragma omp parallel num_threads(threads)
    {
        const size_t workN = (size_t)5 * (TILE_W + 2 * RADIUS) + (size_t)11 * TILE_W + (size_t)5 * BAND_H * TILE_W;
        float* work = (float*)malloc(sizeof(float) * (workN + 16));
        float* ssim = (float*)malloc(sizeof(float) * ((size_t)BAND_H * TILE_W + 16));
        if (!work || !ssim) {
#pragma omp atomic write
            err = 12;
        } else {
#pragma omp for schedule(dynamic, 4)
            for (int64_t t = 0; t < bands * tilesX; ++t) {
                const int64_t bi = t / tilesX, tx = t % tilesX;
                const int64_t y0 = bi * BAND_H, x0 = tx * TILE_W;
                const int64_t bh = (H - y0 < BAND_H) ? H - y0 : BAND_H;
                const int64_t tw = (W - x0 < TILE_W) ? W - x0 : TILE_W;
                tile32(&jb, x0, tw, y0, bh, work, ssim);
                tileSums[t] = tile_sum_avx_order(ssim, tw, tw, bh);
            }
        }
        free(work);
        free(ssim);
    }
    double sum = 0.0;
    for (int64_t t = 0; t < bands * tilesX; ++t)
        sum += tileSums[t]; /* row-major tile order, src/ssim.cpp:1084-1086 */
    free(tileSums);
    if (err)
        return err;
    if (sum_out)
        *sum_out = sum;
    if (ssim_out)
        *ssim_out = (float)(sum / (double)(uint32_t)(width * height)); /* src/ssim.cpp:1102 (uint32 product) */
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * fp64 oracle: the reference's own test oracle naive::compute_ssim<double,uint8_t>
 * (tests/ssim_naive.h:230-339): plain 121-tap gather, taps in raster order, 64x64 tiles for the
 * summation order (tileSum per tile, sum += tileSum, :262/:330), all in double.
 * map is double here (the naive oracle's map type is F).
 * ------------------------------------------------------------------------------------------ */
int oracle_ssim_naive_f64(double* ssim_out, double* sum_out, uint32_t width, uint32_t height,
                          const uint8_t* a, ptrdiff_t a_step, ptrdiff_t a_stride,
                          const uint8_t* b, ptrdiff_t b_step, ptrdiff_t b_stride,
                          double* map, ptrdiff_t map_step, ptrdiff_t map_stride, int threads)
{
    if (!a || !b)
        return 22;
    double k[121];
    oracle_kernel121_f64(k);
    const double c1 = (0.01 * 255.0) * (0.01 * 255.0);
    const double c2 = (0.03 * 255.0) * (0.03 * 255.0);
    const int64_t W = width, H = height;
    const int64_t T = 64;
    const int64_t tilesX = (W + T - 1) / T, tilesY = (H + T - 1) / T;
    double* tileSums = (double*)calloc((size_t)(tilesX * tilesY > 0 ? tilesX * tilesY : 1), sizeof(double));
    if (!tileSums)
        return 12;
    if (threads < 1)
        threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int64_t t = 0; t < tilesX * tilesY; ++t) {
        const int64_t ty = (t / tilesX) * T, tx = (t % tilesX) * T;
        const int64_t th = (H - ty < T) ? H - ty : T, tw = (W - tx < T) ? W - tx : T;
        double tileSum = 0.0;
        for (int64_t y = ty; y < ty + th; ++y) {
            for (int64_t x = tx; x < tx + tw; ++x) {
                double m[5] = {0, 0, 0, 0, 0};
                const double* kp = k;
                for (int64_t ys = y - RADIUS; ys <= y + RADIUS; ++ys) {
                    const int64_t cy = clampi(ys, 0, H - 1);
                    for (int64_t xs = x - RADIUS; xs <= x + RADIUS; ++xs) {
                        const int64_t cx = clampi(xs, 0, W - 1);
                        const double av = (double)a[cx * a_step + cy * a_stride];
                        const double bv = (double)b[cx * b_step + cy * b_stride];
                        const double kv = *kp++;
                        m[0] += kv * av;
                        m[1] += kv * bv;
                        m[2] += kv * (av * av);
                        m[3] += kv * (bv * bv);
                        m[4] += kv * (av * bv);
                    }
                }
                const double muA2 = m[0] * m[0], muB2 = m[1] * m[1], muAB = m[0] * m[1];
                const double sA2 = m[2] - muA2, sB2 = m[3] - muB2, sAB = m[4] - muAB;
                const double num = (2 * muAB + c1) * (2 * sAB + c2);
                const double den = (muA2 + muB2 + c1) * (sA2 + sB2 + c2);
                const double v = num / den;
                tileSum += v;
                if (map)
                    map[y * map_stride + x * map_step] = v;
            }
        }
        tileSums[t] = tileSum;
    }
    double sum = 0.0;
    for (int64_t t = 0; t < tilesX * tilesY; ++t)
        sum += tileSums[t];
    free(tileSums);
    if (sum_out)
        *sum_out = sum;
    if (ssim_out)
        *ssim_out = sum / (double)(uint32_t)(width * height); /* tests/ssim_naive.h:338 */
    return 0;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}
