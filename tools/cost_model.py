#!/usr/bin/env python3
"""Instruction-issue model of the strip kernels' main loops: instruction counts from the compiled ISA x the per-instruction
SIMD time measured at forced occupancy (tools/occupancy_probe.hip, profiles/r04_occupancy_probe.txt), against the measured
kernel time of the bench line (profiles/r04_final_bench.json).  Answers "is the kernel at the limit of its instruction stream?".

usage: tools/cost_model.py kernels.s [bench.json]
       (kernels.s: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Issim_amd/csrc --cuda-device-only -S ssim_amd/csrc/ssim_kernels.hip)
"""
import collections
import json
import re
import sys

# clk of SIMD time per wave-instruction at 2 / 3 waves per SIMD (2.375 GHz)
COST = {
    "pk":       {2: 4.78, 3: 4.41},   # v_pk_fma/add/mul_f32
    "unpacked": {2: 3.10, 3: 2.74},   # v_fma/add/mul/sub_f32 and other full-rate 32-bit VALU
    "cvt":      {2: 4.46, 3: 4.36},   # v_cvt_f32_ubyteN, v_cvt_f64_f32
    "f64":      {2: 4.68, 3: 4.59},   # v_add_f64 (v_fma_f64 taken as the same)
    "rcp":      {2: 8.54, 3: 8.35},
    "lds_read": {2: 4.3,  3: 4.9},    # per instruction, b64 or b128 alike
    "lds_write":{2: 9.0,  3: 8.3},
    "vmem":     {2: 4.5,  3: 4.5},    # issue only (assumed; the probe's loads expose their latency)
}


def classify(op):
    if op.startswith("v_pk_"): return "pk"
    if op.startswith("v_rcp"): return "rcp"
    if op.startswith("v_cvt_f32_ubyte") or op.startswith("v_cvt_f64"): return "cvt"
    if op.endswith("_f64") or "_f64_" in op: return "f64"
    if op.startswith("v_"): return "unpacked"
    if op.startswith("ds_read"): return "lds_read"
    if op.startswith("ds_write"): return "lds_write"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    return None


def main_loop(lines):
    lab = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: lab[m.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        m = re.search(r's_cbranch_scc0\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in lab and lab[m.group(1)] < i:
            n = sum(1 for x in lines[lab[m.group(1)]:i] if 'v_pk_' in x or '_f64' in x)
            if best is None or n > best[0]: best = (n, lab[m.group(1)], i)
    return lines[best[1]:best[2] + 1]


def main():
    s = open(sys.argv[1]).read()
    bench = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
    kernels = [("exact", "ssim_strip2_kernelILi0ELi0ELb0EE", 2, 278), ("fast (hybrid)", "ssim_strip2_kernelILi1ELi0ELb0EE", 2, 220),
               ("separable", "ssim_strip2_kernelILi4ELi0ELb0EE", 3, 119), ("double + map (one column per lane: a row = 64 pixels)", "ssim_strip1_kernelILi2ELb1ELb0EE", 3, 137)]
    measured = {}
    if bench:
        W, H, pairs = bench["config"]["width"], bench["config"]["height"], bench["config"]["pairs_per_gpu"]
        measured = {"exact": bench["roofline"]["kernel_avg_ms"], "fast (hybrid)": bench["fast_mode"]["kernel_avg_ms"], "separable": bench["separable_mode"]["kernel_avg_ms"]}
        dbl = bench.get("configs", {}).get("4k double + map")
        if dbl: measured["double + map (one column per lane: a row = 64 pixels)"] = dbl["kernel_avg_ms"]
    print("| kernel (main loop, per ROW of a wave = 128 pixels) | waves/SIMD | packed | unpacked | cvt | f64 | rcp | LDS reads | LDS writes | VMEM | model clk | of it blur arithmetic | measured clk | measured / model |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, sym, waves, ops in kernels:
        m = re.search(r'^(_ZN8ssim_hip\S*%s\S*):' % sym, s, re.M)
        nxt = re.search(r'^\s*s_endpgm', s[m.end():], re.M)
        body = main_loop(s[m.start():m.end() + nxt.end()].split('\n'))
        c = collections.Counter()
        for l in body:
            t = l.strip()
            if not t or t[0] in ';.' or t.endswith(':'): continue
            k = classify(t.split()[0])
            if k: c[k] += 1
        per_row = {k: v / 2.0 for k, v in c.items()}          # the loop body is two rows
        clk = sum(per_row[k] * COST[k][waves] for k in per_row)
        if name.startswith("double"):
            blur_clk = 88 * COST["f64"][waves]       # the fp64 multiply-adds of the four blur streams + formula per row and lane
        else:
            blur = {"exact": 255, "fast (hybrid)": 197, "separable": 88}[name]     # packed instructions of the blur streams per row and lane (= lane-ops per pixel)
            blur_clk = blur * COST["pk"][waves]
        meas = ""
        ratio = ""
        if name in measured:
            # wave-rows of the launch: strips x (rows + 10 warm-up); strips of 512 rows at these sizes
            strips_y = (H + 511) // 512
            n_pairs, strip_w = (4, 64) if name.startswith("double") else (pairs, 128)      # the fp64 config of the bench line is 4 x 4096^2
            wave_rows = n_pairs * ((W + strip_w - 1) // strip_w) * strips_y * ((H + strips_y - 1) // strips_y + 10)
            mc = measured[name] * 1e-3 * 2.375e9 * 1024 / wave_rows
            meas, ratio = "%.0f" % mc, "%.2f" % (mc / clk)
        print("| %s | %d | %g | %g | %g | %g | %g | %g | %g | %g | %.0f | %.0f (%.0f %%) | %s | %s |" % (
            name, waves, per_row.get("pk", 0), per_row.get("unpacked", 0), per_row.get("cvt", 0), per_row.get("f64", 0), per_row.get("rcp", 0),
            per_row.get("lds_read", 0), per_row.get("lds_write", 0), per_row.get("vmem", 0), clk, blur_clk, 100 * blur_clk / clk, meas, ratio))


if __name__ == "__main__":
    main()
