// ssim_kernels.h -- internal interface between the C ABI (ssim_hip_abi.cpp) and the gfx950
// kernels (ssim_kernels.hip).  Not installed; nothing here is visible through include/rmgr/.
#ifndef SSIM_AMD_KERNELS_H
#define SSIM_AMD_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ssim_hip {

// One image pair as the kernels address it: pixel (x,y) of A is a[x*a_step + y*a_stride]
// (bytes, signed), map element (x,y) is map[x*map_step + y*map_stride] (floats, signed).
// Mirrors rmgr_ssim_Params (include/rmgr/ssim.h) with device pointers.
struct PairDesc {
    const uint8_t* a;  int64_t a_step, a_stride;
    const uint8_t* b;  int64_t b_step, b_stride;
    float*         map; int64_t map_step, map_stride;
};

// MODE_FAST: the three E[.] planes in the reference's exact order, the two mu planes separable (inside north_star's
// FMA-relative tolerance on all of the reference's test sets); MODE_SEPARABLE: everything separable, four planes, centred
// pixels (closer to the exact value than the reference itself, but not correlated with the reference's rounding).
enum Mode { MODE_EXACT = 0, MODE_FAST = 1, MODE_DOUBLE = 2, MODE_UNFUSED = 3, MODE_SEPARABLE = 4 };

struct Geometry {
    uint32_t width, height, count;
    uint32_t strip_w;      // output columns per wavefront strip (64 or 128, by kernel)
    uint32_t strip_rows;   // output rows per wavefront strip
    uint32_t strips_x, strips_y;  // strips of THIS launch (its row window)
    uint32_t y_begin, y_end;      // output rows [y_begin, y_end) this launch produces; y_begin is a multiple of cell_rows
    uint32_t cell_rows;           // rows per reduction cell: cell_rows_for(height)
    uint32_t cells_x, cells_y;    // the image's 64-column x cell_rows-row reduction cells (ssim_kernels.hip, cell_flush*)
    bool     map_unit;            // every pair of the launch writes its map with ssimStep == 1 (set by the caller of plan(); scheduling only)
    bool     wide;                // some pair needs the fully 64-bit form of the one-column kernel (!fits_strip2(); set by the caller of plan())
    uint32_t wave_slots;          // strips the chip holds at a time with this kernel (SIMDs x waves per SIMD): plan()'s packing unit
    uint32_t chunk_cells;         // > 0: the balanced schedule of the two-column kernel -- every wavefront covers this many cell rows of the launch's
    uint32_t n_chunks;            //      flattened [image][strip column][cell row] list (n_chunks wavefronts); 0: one strip per wavefront
    uint32_t bal_stride;          // balanced schedule: images interleaved column by column in the list (1: none); ssim_kernels.hip work_setup()
    uint32_t xcds;                // XCDs of the device (hipDeviceAttributeNumberOfXccs): the workgroup renumbering's modulus
    uint32_t partials_per_image() const { return cells_x * cells_y; }
};

// The two-column kernel addresses a strip's pixels as (64-bit uniform row base) + (32-bit lane offset) and keeps
// its coordinates in 32-bit registers; a pair outside these (enormous) ranges runs on the one-column kernel,
// which is fully 64-bit: pass variant 1 to plan()/launch() when this returns false.
inline bool fits_strip2(const PairDesc& d, uint32_t width, uint32_t height)
{
    const int64_t lim = int64_t(1) << 21;     // 144 columns x |step| x 4 B (map) stays below 2^31
    return width < 0x7FFF0000u && height < 0x7FFF0000u &&
           d.a_step > -lim && d.a_step < lim && d.b_step > -lim && d.b_step < lim &&
           (d.map == 0 || (d.map_step > -lim && d.map_step < lim));
}

// 2..4 when the batch is made of runs of that many descriptors whose images are the byte-interleaved channels
// of one pair (same steps/strides, base pointers one byte apart, as rmgr_ssim_init_interleaved produces for
// channelNum 0,1,2..); 1 otherwise.
inline int interleaved_group(const PairDesc* d, uint32_t count)
{
    for (int g = 4; g >= 2; --g) {
        if (count < (uint32_t)g || count % (uint32_t)g != 0) continue;
        bool ok = true;
        for (uint32_t i = 0; i < count && ok; ++i) {
            const PairDesc& h = d[i - i % g];
            const int64_t m = i % g;
            ok = d[i].a == h.a + m && d[i].b == h.b + m && d[i].a_step == h.a_step && d[i].b_step == h.b_step &&
                 d[i].a_stride == h.a_stride && d[i].b_stride == h.b_stride && h.a_step >= g && h.b_step >= g;
        }
        if (ok) return g;
    }
    return 1;
}

// Strip geometry the launcher will use for (mode, variant, requested rows; 0 = default).
// The library default (tuning variant 0) is the two-column kernel, except for launches too small to give even a
// quarter of the SIMDs a strip at the minimum strip height: there the one-column kernel's twice-as-many, half-as-wide
// strips finish sooner (one 256^2 or 512^2 pair, bit-exact modes and the hybrid: 19.3 -> 15.4 us; equal from 1024^2 on;
// the three-wave kernels gain nothing).  Results do not depend on the choice (the cell reduction is common to both kernels).
inline int default_variant(uint32_t width, uint32_t height, uint32_t count, int mode, int cu_count)
{
    if (mode != MODE_EXACT && mode != MODE_UNFUSED && mode != MODE_FAST) return 0;     // the two-waves-per-SIMD kernels
    const uint64_t strips = (uint64_t)((width + 127) / 128) * ((height + 7) / 8) * count;
    return strips <= (uint64_t)(cu_count > 0 ? cu_count : 256) ? 1 : 0;      // fewer strips than CUs (a quarter of the SIMDs)
}

// Rows per cell of the fp64 reduction: a function of the image height ONLY, so that every launch that touches an
// image of this size -- any strip height, batch, band or GPU -- builds the same cells.  Images below 2048 rows keep
// 8-row cells: a lone small image is cut into 8-row strips to fill the GPU, and 1080 rows split into five even
// 216-row strips only in 8-row units (16-row cells -> 4 x 224 + 184 measured 2 % slower than the cells save).
// Taller images amortise the per-cell cost over 32 rows (MODE_EXACT, 32 x 4096^2: -0.8 % vs no cells at all, 8-row
// cells -1.5...-3 %; profiles/r02_cells_ab.txt).
inline uint32_t cell_rows_for(uint32_t height) { return height >= 2048 ? 32u : 8u; }

// Tuning variants that force the balanced schedule of the two-column kernel: 6 with plan()'s interleave of the images in the chunk list, 7 without
// any (round 5's list), 100 + T with T images interleaved (measurement aids; ssim_kernels.hip work_setup()).
enum { kClockStride = 5, kClockMaxXcds = 16, kClockWords = kClockStride * kClockMaxXcds };

inline bool is_balanced_variant(int variant) { return variant == 6 || variant == 7 || variant >= 100; }

// Which launches run the EARLY form of the bit-exact two-column kernel by default.  Measured with both forms interleaved in
// one process over batch sizes (profiles/r03_early_sweep.txt; rounds = strips / wave slots of the chip): up to ~2 rounds EARLY
// wins 2.4...5 % (one 4096^2 pair 169.5 -> 173.5 Gpix/s, 8 x 4096^2 201 -> 206, 2 x 8192^2 + map 190 -> 200, 32 x 1080p
// 186 -> 193), around 3 rounds +1 %, at 4 rounds +-0.4 %, and on longer launches it LOSES (192 / 256 / 384 x 1080p -3 / -4 /
// -2.5 %; 64...128 x 4096^2 -0...0.5 %).  Why the sign flips with the launch length is not understood (more VALU work in the
// low-priority phase; the waves of long launches start staggered, those of one-round launches in lockstep), so the rule is
// the measured one.
inline bool uses_early_row_sums(const Geometry& geo, int mode, int variant)
{
    if ((mode != MODE_EXACT && mode != MODE_UNFUSED) || variant == 1 || geo.strip_w != 128) return false;
    if (variant == 3 || is_balanced_variant(variant)) return true;      // the balanced schedule exists with EARLY row sums only
    if (variant != 0) return false;
    const uint64_t strips = (uint64_t)geo.strips_x * geo.strips_y * geo.count;
    return geo.wave_slots != 0 && strips <= 3ull * geo.wave_slots;
}

// y_begin / y_rows: the output rows the launch produces (default: the whole image).  A host that pipelines an image
// in row bands launches consecutive windows -- each starting on a cell boundary -- into the same partials and
// asks for the reduction with the last one; the sums are bit-identical to the single launch's.
// cu_count / xcd_count: the device's (hipDeviceProp_t::multiProcessorCount, hipDeviceAttributeNumberOfXccs); <= 0: MI355X's 256 / 8.
Geometry plan(uint32_t width, uint32_t height, uint32_t count, int mode, int strip_rows, int variant, int cu_count, int xcd_count,
              uint32_t y_begin = 0, uint32_t y_rows = 0xFFFFFFFFu);

// Doubles of device scratch launch() needs for `geo` (cell partials + the chunk sums of the two-stage reduction).
size_t partials_size(const Geometry& geo);

// Enqueues the SSIM kernel + the per-image reduction on `stream`.
//   descs_dev   count descriptors in device memory, or NULL when count == 1 and `single` is used
//   partials    device scratch, >= partials_size(geo) doubles
//   reduce      false: only the strip kernel of this row window runs (more windows follow); true: + the reduction
//   sums        device, count doubles: per-image fp64 sum of the SSIM values
//   group       > 1 when every run of `group` consecutive descriptors addresses the interleaved channels of one
//               image pair (interleaved_group()): scheduling hint only, results do not depend on it
// ev_begin/ev_end (optional) are recorded around the main kernel only.
// clock (optional, profiling): kClockWords device uint64 (5 per XCD: start cycles, start ticks, sum of cycles, sum of reference ticks, launches); the first geo.xcds
// workgroups of the strip kernel -- one per XCD -- add their own run to them.
hipError_t launch(const Geometry& geo, int mode, int variant, int group, const PairDesc* descs_dev, const PairDesc& single,
                  double* partials, double* sums, hipStream_t stream, hipEvent_t ev_begin, hipEvent_t ev_end, bool reduce = true, uint64_t* clock = nullptr);

// The reduction alone: per-image sums of `geo.count` images' cell partials (partials: [image][cell_y][cell_x], the layout
// launch() writes), in the fixed order every launch uses.  chunk_sums: reduce_scratch_size(geo) doubles of device scratch
// (0 for images of up to 8192 cells).  For hosts that assemble an image's cells from several row-band launches -- possibly
// on several GPUs -- before reducing them.
size_t reduce_scratch_size(const Geometry& geo);
hipError_t launch_reduce(const Geometry& geo, const double* partials, double* chunk_sums, double* sums, hipStream_t stream);

// sha256 of ssim_kernels.hip as it was compiled (the Makefile passes it in; "unknown" for any other build).
const char* kernels_source_id();

// BT.601 luminance of interleaved pixels (src/ssim-cli.cpp:158-186), device to device.
hipError_t launch_luminance(uint8_t* dst, int64_t dst_stride, const uint8_t* src, int64_t src_step, int64_t src_stride,
                            uint32_t width, uint32_t height, hipStream_t stream);

// The synthetic test pattern of SURVEY.md 8(d) (integer-only, bit-reproducible; ssim_amd/synth.py and
// oracle_synth_pair are its host twins), written straight into device memory: A and B planes, rows stride bytes apart.
hipError_t launch_synth_pair(uint8_t* a, int64_t a_stride, uint8_t* b, int64_t b_stride, uint32_t width, uint32_t height,
                             uint64_t seed, hipStream_t stream);

// Profiling aid (ssim_probe.hip): a pure packed-fp32 stream at a FORCED occupancy of waves_per_simd (1, 2, 3, 4 or 8) waves per SIMD on
// a grid of exactly cu_count x 4 x waves_per_simd single-wave workgroups; stream_kind 0: independent v_pk_fma_f32, 1: two interleaved
// dependent chains of six (the shape of the blur's row sums).  probe_valu_lane_ops(): the lane-operations one such launch retires.
hipError_t launch_probe_valu(int waves_per_simd, int stream_kind, int cu_count, int xcd_count, int iters, float* out, hipStream_t stream, uint64_t* clock = nullptr);   // clock: as launch()'s
uint64_t probe_valu_lane_ops(int waves_per_simd, int cu_count, int iters);

} // namespace ssim_hip

#endif
