#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: Mpix/s of compute_ssim (global SSIM,
no map) on 4096x4096 uint8 pairs, with the achieved-vs-roofline figures and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--workload ...]

`--gpus N` with N > 1 is self-contained, like the reference's rmgr_ssim_compute_ssim_openmp builds its
own pool (src/ssim-openmp.c:40-47): when no launcher has set WORLD_SIZE, this process -- before it
imports torch or touches HIP -- starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 ... bench.py <same flags>` as a CHILD process (never exec), relays its output
(rank 0 prints the JSON line) and exits with its code.  Launched under torch.distributed.run by someone
else (RANK / LOCAL_RANK / WORLD_SIZE set) it is simply one of the ranks.

A "step" is one pass of the hot path over one batch: distinct synthetic pairs (seeds 0x5EED+i,
SURVEY.md 8(d)) resident in HBM, one batched launch per rank through the C ABI
(rmgr_ssim_hip_enqueue_batch), and -- for N > 1 -- one RCCL all-reduce of the per-image fp64 sums so
that every rank holds every result.  That all-reduce is the PRODUCT's (`--exchange native`:
rmgr_ssim_hip_comm_allreduce_sums behind the C ABI; torch.distributed only launches the ranks, hands
rank 0's communicator id around and takes the max of the elapsed times) whenever its communicator comes
up on every rank within its deadline -- the default, `auto`, otherwise falls back to
torch.distributed.all_reduce and says so in `exchange` (carrier, ranks RCCL counted, all-reduce time per
step, and that the other carrier delivers the same bits).  Images are sharded by rank:
    --scaling weak   (default) every rank owns PAIRS pairs: per-GPU work is fixed;
    --scaling strong the workload's total batch (e.g. BASELINE.json configs[3]: 1024 x 1080p) is split
                     over the ranks with sharding.split_batch: total work is fixed.
Timing: barrier + synchronize on both sides of exactly K steps, max over ranks; value = all pixels of
all ranks / that time.  Before any timing the results are gated on the reference's known answers
(FMA path float bits) -- on every rank's first pairs.

After the headline a single-GPU run times the other BASELINE.json configs (8192^2 + map exact
and separable, 128 x 1080p, fp64 internals 4096^2 + map), each KAT-gated, into `configs`.

Only the cpu_baseline leg touches oracle/: it times the real reference kernels (oracle/_ref,
kind "reference") or, where that prebuilt library is absent, the C restatement (kind "port").
"""
import argparse
import ctypes
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs as selectable workloads; the default ("4k") is the one the metric is quoted on.
#   name: (width, height, pairs per GPU (weak), total pairs (strong), write map, KATs of the first pairs =
#          reference FMA float bits for seeds 0x5EED, 0x5EEE, ..., description)
WORKLOADS = {
    "4k":     (4096, 4096, 32, 256, False, (0x3f64b7be,), "4096x4096 uint8 pairs (BASELINE.json configs[1] image), global SSIM only"),
    "8k-map": (8192, 8192, 2, 16, True, (0x3f64b5b4,), "8192x8192 uint8 pairs with per-pixel SSIM map writeback (BASELINE.json configs[2])"),
    "1080p":  (1920, 1080, 128, 1024, False, (0x3f64bb1f, 0x3f64bbf6, 0x3f64bb30),
               "1920x1080 uint8 pairs, global SSIM only (BASELINE.json configs[3]: 1024 pairs over 8 GPUs = 128 per GPU)"),
}
NAIVE_F64_4K = 0.893428737869049      # tests/ssim_naive.h compute_ssim<double> on the 4096^2 seed-0x5EED pair
HBM_PEAK_GBS = 8000.0                 # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# VALU work per output pixel, lane-ops: exact = 5 planes x (5 fold adds + 6 mul + 30 fma + 10 ring adds) = 255
# + 23 for the SSIM formula / divide / fp64 accumulate; fast (hybrid) = 3 reference-order planes x 51 + 2 separable planes
# x 22 + 23; separable = 4 planes x 22 + 23 + 8 (a^2 + b^2 is blurred as one plane; centring and the restored mu);
# fp64 mode: VALU ISSUE SLOTS per pixel counted in the compiled hot loop (tools/isa_mix.py: 274 VALU instructions per two
# rows of one pixel per lane) -- 88 fp64 multiply-adds of the blur and the formula, 25 fp32->fp64 conversions of the folded
# sums, 10 packed fold adds, 14 of staging / in-range division / map value; every non-packed instruction
# occupies one fp64-rate slot, which is what the 39.3 T/s peak counts (round 2 counted the 88 blur operations only)
VALU_OPS_PER_PIXEL = {0: 278, 1: 220, 2: 137, 3: 278, 4: 119}
FP64_MATH_OPS_PER_PIXEL = 88          # mode 2 only: the fp64 multiply-adds of blur + formula among those 137 slots (round 2's accounting; kept so that rounds compare)
VALU_PEAK_TOPS = 78.6                 # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz lane-ops/s; = 157.3 TFLOP/s fp32 vector spec / 2
VALU_PEAK_F64_TOPS = 39.3             # fp64 vector: 78.6 TFLOP/s spec / 2
# Rounds 4-5 divided every kernel by two CONSTANTS measured once, on one round-4 box (tools/occupancy_probe.hip: 74.6 / 65.1 T lane-ops/s at 8 / 2 waves
# per SIMD) -- while the boxes of the pool differ by +-4 %.  Since round 6 the line carries the peak of the box it ran on: rmgr_ssim_hip_probe_valu (the
# same forced-occupancy v_pk_fma_f32 stream, inside the library) runs in-process before the clock-settle loop and right after the timed steps; the
# constants below are kept in the line for comparison with those rounds only, no fraction is computed from them any more.
ROUND4_BOX_VALU_TOPS = {"8wave": 74.6, "2wave": 65.1, "3wave": 70.7}
KERNEL_WAVES_PER_SIMD = {0: 2, 1: 2, 2: 3, 3: 2, 4: 3}      # what each mode's strip kernel runs at (its VGPR count; ssim_kernels.hip waves_per_simd())
MODE_NAMES = ["exact (reference FMA order, bit-faithful)", "fast (reference-order E planes + separable mu planes; inside the FMA-relative tolerance on the reference's image sets, not a guarantee)",
              "double (fp64 internals)", "unfused (reference AVX order)", "separable (all planes separable fp32, four planes, centred; reference test tolerance vs the exact value)"]
# tests/ssim_naive.h<double> known answers of the synthetic pairs, seeds 0x5EED, 0x5EEE, ...
NAIVE_KATS = {(4096, 4096): (0.893428737869049,), (8192, 8192): (0.893397634039865,),
              (1920, 1080): (0.893480304106111, 0.893493105227394, 0.893481282347413)}


# ------------------------------------------------------------------------------------------------
# self-launch (no torch, no HIP before this returns)
# ------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(n, argv, port=None):
    """The torch.distributed.run command line that starts `n` ranks of this script with `argv`."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()),
            os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["SSIM_BENCH_CHILD"] = "1"
    cmd = launcher_command(n, argv)
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------
# CPU baseline (the only user of oracle/)
# ------------------------------------------------------------------------------------------------
def host_cpu():
    model, procs = "unknown", os.cpu_count()
    try:
        with open("/proc/cpuinfo") as f:
            for l in f:
                if l.startswith("model name"):
                    model = l.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, procs


CPU_BASELINE_ENV = {"OMP_PROC_BIND": "close", "OMP_PLACES": "cores", "OMP_DYNAMIC": "false"}


def cpu_baseline_in_child(budget_s=10.0):
    """Runs cpu_baseline() in a CHILD process (never exec: this process has initialised the GPU) whose OpenMP runtime starts
    with its threads pinned -- OMP_PROC_BIND=close, OMP_PLACES=cores: one thread per physical core, neighbouring cores
    first -- because the OpenMP runtime reads those variables once, when it is loaded, and this process loaded one long
    ago (torch).  Unpinned, the 64-thread baseline varied 1.9x between its best and its median run (round 3)."""
    env = dict(os.environ)
    env.update(CPU_BASELINE_ENV)
    env.pop("OMP_NUM_THREADS", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", str(budget_s)], env=env, stdout=subprocess.PIPE, timeout=600)
    if r.returncode != 0:
        raise SystemExit("the cpu_baseline child failed (exit %d)" % r.returncode)
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


def cpu_baseline(budget_s=10.0):
    """Reference FMA+OpenMP path (or the port) on this box's host cores.  Headline: one 4096^2 pair, global only,
    all threads (the reference caps its pool at 64, src/ssim.cpp:1025), repeated for ~budget_s; plus, per BASELINE
    config, one thread and all threads (the table the reference's own test binary prints, tests/rmgr-ssim-tests.cpp:188-222)."""
    import numpy as np
    import oracle
    if oracle.have_ref():
        kind = "reference"
        cores = min(oracle.ref_lib().ref_max_threads(), 64)
        run = lambda a, b, threads, want_map=False, out_map=None: oracle.ref_ssim(a, b, want_map=want_map, impl=5, threads=threads, out_map=out_map)
    else:
        kind = "port"
        cores = oracle.oracle_lib().oracle_max_threads()
        run = lambda a, b, threads, want_map=False, out_map=None: oracle.ssim_f32(a, b, want_map=want_map, fused=True, threads=threads, out_map=out_map)

    def timed(fn, budget, min_runs=3, max_runs=100000):
        ts = []
        stop = time.perf_counter() + budget
        while (time.perf_counter() < stop or len(ts) < min_runs) and len(ts) < max_runs:
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return ts

    W, H = 4096, 4096
    a, b = oracle.synth_pair(W, H, 0x5EED)
    v = run(a, b, cores)[0]
    assert int(np.float32(v).view(np.uint32)) == WORKLOADS["4k"][5][0], "CPU baseline disagrees with the known answer"
    ts = timed(lambda: run(a, b, cores), budget_s)
    t1 = timed(lambda: run(a, b, 1), 0.0, min_runs=2)
    px = W * H
    per = {"4k": {"pixels": px, "map": False, "threads_all_mpix_s": round(px / min(ts) / 1e6, 1),
                  "threads_all_median_mpix_s": round(px / statistics.median(ts) / 1e6, 1),
                  "one_thread_mpix_s": round(px / min(t1) / 1e6, 1)}}
    # 8192^2 with map (configs[2]) and one 1080p pair (configs[3]'s image)
    for name, (w, h, want_map, kat) in (("8k-map", (8192, 8192, True, WORKLOADS["8k-map"][5][0])),
                                        ("1080p", (1920, 1080, False, WORKLOADS["1080p"][5][0]))):
        aa, bb = oracle.synth_pair(w, h, 0x5EED)
        mm = np.zeros((h, w), np.float32) if want_map else None      # one caller-owned map, touched once: as the reference's harness reuses its buffers
        r = run(aa, bb, cores, want_map, mm)
        assert int(np.float32(r[0]).view(np.uint32)) == kat, "CPU baseline disagrees with the known answer (%s)" % name
        ta = timed(lambda: run(aa, bb, cores, want_map, mm), 2.0)
        to = timed(lambda: run(aa, bb, 1, want_map, mm), 0.0, min_runs=2 if name == "1080p" else 1)
        per[name] = {"pixels": w * h, "map": want_map, "threads_all_mpix_s": round(w * h / min(ta) / 1e6, 1),
                     "threads_all_median_mpix_s": round(w * h / statistics.median(ta) / 1e6, 1),
                     "one_thread_mpix_s": round(w * h / min(to) / 1e6, 1)}
        del aa, bb, r, mm
    # configs[3] as BASELINE.md section 3 defines its CPU figure: the 1024 x 1080p batch looped SERIALLY over pairs with OpenMP inside
    # each call (the reference's harness shape, tests/rmgr-ssim-tests.cpp:293-303).  32 distinct pairs (seeds 0x5EED + i) are timed,
    # three times, the best pass counts; the rate is per pixel, so the 1024-pair figure is the same number (stated, not measured, for pairs 33..1024).
    batch_pairs = [oracle.synth_pair(1920, 1080, 0x5EED + i) for i in range(32)]
    def batch_pass():
        for aa, bb in batch_pairs:
            run(aa, bb, cores)
    batch_pass()
    tb = timed(batch_pass, 0.0, min_runs=3)          # best of three passes (one collection of ten read 0.74 k here where the others read 5.7...6.5 k: a host hiccup during both of two passes)
    per["1080p-batch"] = {"pixels": 32 * 1920 * 1080, "map": False, "pairs_timed": 32, "pairs_in_config": 1024, "loop": "serial over pairs, OpenMP inside each call",
                          "threads_all_mpix_s": round(32 * 1920 * 1080 / min(tb) / 1e6, 1),
                          "seconds_for_1024_pairs_extrapolated": round(min(tb) * 32, 2)}
    del batch_pairs
    # configs[4]: the reference's RMGR_SSIM_USE_DOUBLE build (oracle/_ref/libssim_ref_double.so: the same TUs with Float = double), 4096^2 with and
    # without the map; with a map that build takes its generic sum_tile (src/ssim.cpp:947-950), restated in oracle/ref_harness.cpp
    if kind == "reference" and oracle.have_ref_double():
        rund = lambda threads, mm=None: oracle.ref_ssim(a, b, impl=5, threads=threads, out_map=mm, double=True)
        assert int(np.float32(rund(cores)[0]).view(np.uint32)) == WORKLOADS["4k"][5][0], "CPU baseline (double build) disagrees with the known answer"
        mm = np.zeros((H, W), np.float32)
        td, tdm = timed(lambda: rund(cores), 1.5), timed(lambda: rund(cores, mm), 1.5)
        td1 = timed(lambda: rund(1), 0.0, min_runs=1)
        per["4k-double"] = {"pixels": px, "build": "RMGR_SSIM_USE_DOUBLE=1 (oracle/_ref/libssim_ref_double.so)",
                            "threads_all_mpix_s": round(px / min(td) / 1e6, 1), "threads_all_median_mpix_s": round(px / statistics.median(td) / 1e6, 1),
                            "with_map_threads_all_mpix_s": round(px / min(tdm) / 1e6, 1), "with_map_threads_all_median_mpix_s": round(px / statistics.median(tdm) / 1e6, 1),
                            "one_thread_mpix_s": round(px / min(td1) / 1e6, 1)}
        del mm
    model, procs = host_cpu()
    pinning = ", ".join("%s=%s" % (k, os.environ[k]) for k in sorted(CPU_BASELINE_ENV) if os.environ.get(k)) or "unpinned"
    return {"value": round(px / min(ts) / 1e6, 1), "median": round(px / statistics.median(ts) / 1e6, 1), "unit": "Mpix/s",
            "cores": cores, "threads": cores, "pinning": pinning, "best_over_median": round(statistics.median(ts) / min(ts), 3),
            "kind": kind, "cpu_model": model, "logical_cpus": procs, "runs": len(ts),
            "one_thread_mpix_s": per["4k"]["one_thread_mpix_s"], "per_config": per,
            "sample": "%d back-to-back runs of one 4096x4096 pair (seed 0x5EED, global only) over ~%.0f s on %d threads (%s): value = best run, "
                      "median = median run (the figure to quote); per_config: best of ~2 s (all threads) / of 1-2 runs (1 thread) on one pair of each image size; %s"
                      % (len(ts), budget_s, cores, pinning,
                         "the reference's real FMA/AVX kernel objects (oracle/_ref: src/ssim_fma.cpp, src/ssim_avx.cpp compiled as they are) driven by "
                         "the tile loop of oracle/ref_harness.cpp -- NOT by src/ssim.cpp, which needs a cmake-generated header -- scratch on the executing thread's stack "
                         "and OpenMP's default schedule over tiles as the reference's default path; in the build container this leg runs at the cmake-built reference's "
                         "speed to within that container's run-to-run spread (BASELINE.md 2b: 79-92 / 530-558 Mpix/s at 1 / 8 threads on 4096^2 against 83.1 / 532: "
                         "-5...+11 % per row at one thread, best runs equal at eight, medians 20 % lower); per_config['1080p-batch'] is configs[3] as BASELINE.md "
                         "section 3 defines it (pairs looped serially, OpenMP inside each call; 32 of the 1024 pairs timed), per_config['4k-double'] the reference's "
                         "RMGR_SSIM_USE_DOUBLE build (configs[4])"
                         if kind == "reference" else "oracle/ssim_oracle.c restatement, OpenMP")}


class PowerSampler(object):
    """`rocm-smi --showpower --showclocks --showmaxpower --showtemp` every ~0.2 s in a side thread while the sustained leg runs: socket power against its cap, the shader clock
    and the junction temperature the firmware reports.  Round 6 found the strip kernels POWER-bound -- 1360...1370 W of a 1400 W cap at ~2.2 GHz, where a pure FMA stream draws
    1120 W at 2.36 GHz (tools/power_probe.py) -- and the line should say so on whatever box it runs.  Absent or unreadable rocm-smi: no `package` object, nothing else changes."""

    def __init__(self):
        import threading
        self.samples, self.stop = [], False
        self.thread = threading.Thread(target=self.run, daemon=True)

    def start(self):
        self.thread.start()

    def run(self):
        while not self.stop:
            try:
                r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--showtemp", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=10)
                d = json.loads(r.stdout.decode())
                self.samples.append(d[sorted(d)[0]])
            except Exception:  # noqa: BLE001 -- a probe must not cost the bench its line
                return
            time.sleep(0.2)

    def finish(self):
        self.stop = True
        self.thread.join(timeout=15)

    def summary(self):
        def series(needle):
            out = []
            for smp in self.samples[1:]:                 # the first sample may predate the load
                for k, v in smp.items():
                    if needle in k.lower():
                        try:
                            out.append(float(str(v).strip("()").lower().replace("mhz", "")))
                        except ValueError:
                            pass
                        break
            return sorted(out)
        pw, cap, clk, tj = series("current socket graphics package power"), series("max graphics package power"), series("sclk clock speed"), series("sensor junction")
        if not pw:
            return {}
        med = lambda v: v[len(v) // 2] if v else None
        out = {"samples": len(pw), "power_w_median": med(pw), "power_w_max": pw[-1], "power_cap_w": med(cap), "sclk_mhz_median": med(clk), "junction_c_max": tj[-1] if tj else None,
               "note": "rocm-smi sampled every ~0.2 s during the sustained leg (card 0): socket power against the package's cap, the firmware's shader clock, junction temperature"}
        if med(cap):
            out["power_frac_of_cap"] = round(med(pw) / med(cap), 4)
        return out


def measured_traffic(mode, workload, pairs, kernel_id):
    """(HBM bytes per launch, note) from the committed PMC measurement (profiles/traffic.json: rocprofv3 --pmc passes of
    tools/profile_target.py, FETCH_SIZE/WRITE_SIZE corrected per profiles/r01_fetch_size_calibration.md).  The file names the kernel
    source it was collected from (sha256 of ssim_kernels.hip); the RUNNING library reports the source it was compiled from
    (rmgr_ssim_hip_get_kernel_source_id): for any other kernel version the figure is None -- a measurement of other code is not
    this run's traffic.  The headline batch (32 x 4096^2) is measured as such, nothing is scaled; other batch sizes scale per pair."""
    key = {"4k": "exact_4096_nomap", "8k-map": "exact_8192_map", "1080p": "exact_1080p_nomap"}.get(workload)
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/traffic.json missing or unreadable"
    if mode != 0 or key not in t:
        return None, "no PMC measurement of this configuration in profiles/traffic.json"
    if t.get("kernel_source_sha256") != kernel_id:
        return None, ("profiles/traffic.json was collected from kernel source %s..., the running library was compiled from %s...: not quoted (re-run "
                      "tools/collect_profiles.sh + tools/publish_profiles.sh)" % (str(t.get("kernel_source_sha256"))[:12], kernel_id[:12]))
    e = t[key]
    how = "measured on this very batch" if int(e["pairs"]) == pairs else "measured on %d pairs, scaled per pair to %d" % (e["pairs"], pairs)
    return float(e["bytes_per_pair"]) * pairs, ("HBM bytes per launch from rocprofv3 --pmc passes (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections) of the same kernel "
                                                "source (sha256 %s...), committed as profiles/traffic.json; %s" % (kernel_id[:12], how))


def attainable_hbm_gbs(torch, dev):
    """What a plain device-to-device copy reaches on this box (SURVEY 8d asks for the attainable figure next
    to the 8 TB/s spec): 1 GiB read + 1 GiB written per copy, HIP events, best of 10."""
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev).fill_(3)
    dst = torch.empty_like(src)
    best = None
    for _ in range(3):
        dst.copy_(src)
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    del src, dst
    return round(2.0 * n / (best * 1e-3) / 1e9, 1)


def kernel_name(mode, variant, want_map, plan=None):
    """Template instance rocprofv3 lists for this configuration (ssim_kernels.hip): the two-column kernel's second argument
    is 0 = no map, 2 = map with 8-byte stores (bench maps are dense, widths even), its third whether the bit-exact modes run their
    EARLY form (short launches), its fourth the balanced schedule (bit-exact modes and MODE_FAST, no map; both from rmgr_ssim_hip_get_plan:
    `plan`); the one-column kernel's are bools (map, 64-bit addressing)."""
    if mode == 2 or variant == 1:
        return "ssim_strip1_kernel<%d, %s, false>" % (mode, "true" if want_map else "false")     # last argument: 64-bit addressing (never needed by the bench's pairs)
    balanced = bool(plan is not None and plan.balancedChunks and not want_map and mode in (0, 3, 1))
    early = (balanced and mode != 1) or bool(plan is not None and plan.earlyRowSums)
    return "ssim_strip2_kernel<%d, %d, %s, %s>" % (mode, 2 if want_map else 0, "true" if early else "false", "true" if balanced else "false")


def figures(mode, pairs, w, h, want_map, kernel_avg_ms):
    """roofline / valu objects of one launch of `pairs` w x h pairs that took kernel_avg_ms."""
    bpp = 6 if want_map else 2                    # algorithmic HBM bytes per pixel pair
    px = float(pairs) * w * h
    sec = kernel_avg_ms * 1e-3
    gbs = px * bpp / sec / 1e9
    ops = VALU_OPS_PER_PIXEL[mode]
    peak = VALU_PEAK_F64_TOPS if mode == 2 else VALU_PEAK_TOPS
    t = ops * px / sec / 1e12
    valu = {"achieved": round(t, 2), "peak": peak, "unit": "T fp64-rate issue slots/s" if mode == 2 else "T lane-ops/s",
            "frac": round(t / peak, 4), "ops_per_pixel": ops}
    if mode == 2:     # both accountings: issue slots of every kind against the fp64-rate issue peak, and fp64 arithmetic alone
        tm = FP64_MATH_OPS_PER_PIXEL * px / sec / 1e12
        valu.update({"fp64_math_ops_per_pixel": FP64_MATH_OPS_PER_PIXEL, "fp64_math_achieved": round(tm, 2), "fp64_math_frac": round(tm / peak, 4),
                     "note": "frac counts all 137 VALU issue slots per pixel (conversions, staging, division included; not all are fp64 arithmetic); "
                             "fp64_math_frac counts the 88 fp64 multiply-adds only -- the figure comparable with round 2"})
    return ({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
             "algorithmic_bytes_per_launch": px * bpp}, valu)


class Batch(object):
    """`count` resident synthetic pairs (global seeds first_seed_index + i) and their C-ABI parameter blocks."""

    def __init__(self, torch, ssim_amd, synth, ctx, dev, w, h, first, count, want_map):
        self.w, self.h, self.count, self.want_map = w, h, count, want_map
        self.imgs = []
        self.params = (ssim_amd.Params * max(count, 1))()
        for i in range(count):
            # torch owns the memory; the library's own generator (rmgr_ssim_hip_synth_pair_device) fills it on the
            # context's stream, which is torch's current stream
            a = torch.empty((h, w), dtype=torch.uint8, device=dev)
            b = torch.empty((h, w), dtype=torch.uint8, device=dev)
            ctx.synth_pair(a.data_ptr(), w, b.data_ptr(), w, w, h, synth.BASE_SEED + first + i)
            m = torch.empty((h, w), dtype=torch.float32, device=dev) if want_map else None
            self.imgs.append((a, b, m))
            self.params[i] = ssim_amd.make_params(w, h, a.data_ptr(), 1, w, b.data_ptr(), 1, w, m.data_ptr() if want_map else None, 1, w)


PEAK_LANE_OPS_PER_CLOCK = 32768.0      # 256 CUs x 4 SIMDs x 16 lanes x 2 (packed): the data sheet's 78.6 T lane-ops/s is this at 2.4 GHz


def probe_box(ctx, occupancies=(2, 8)):
    """{waves per SIMD: (T lane-ops/s, shader MHz)} a pure v_pk_fma_f32 stream sustains on this box right now at each forced occupancy (rmgr_ssim_hip_probe_valu:
    three bursts of untimed + five timed ~2 ms launches per occupancy, the best burst's median; the clocks are what one workgroup per XCD of that burst measured)."""
    out = {}
    for w in occupancies:
        t, mhz, lo = ctx.probe_valu(int(w), 0, 5, with_clock=True)
        out[int(w)] = (round(t, 2), round(mhz, 1), round(lo, 1))
    return out


def probe_box_or_none(ctx, occupancies):
    """probe_box(), or None when the profiling aid fails (the yardstick is an extra: the bench line must not depend on it)."""
    try:
        return probe_box(ctx, occupancies)
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("bench.py: rmgr_ssim_hip_probe_valu failed (%s): no box-relative fractions in this line\n" % e)
        return None


def against_box(valu, mode, samples, kernel_clock=None):
    """Adds the box-relative fractions to a `valu` object: `samples` = probe_box() results taken around the timed launches (mean of them per occupancy);
    kernel_clock = (mean, slowest XCD's) shader MHz the timed launches ran at (rmgr_ssim_hip_get_profile_clock)."""
    if mode == 2 or not samples:
        return valu     # fp64 internals: its unit is fp64-rate issue slots, the packed-fp32 stream is not its yardstick
    waves = KERNEL_WAVES_PER_SIMD[mode]
    best = lambda w: max(samples, key=lambda smp: smp[w][0])[w]        # the peak is the best sample: a probe burst can run degraded (rmgr/ssim-hip.h), never enhanced
    mean = lambda w, i: best(w)[i]
    at_kernel, at_8 = best(waves)[0], best(8)[0]
    valu.update({"kernel_waves_per_simd": waves, "box_peak_%dwave" % waves: round(at_kernel, 2), "box_peak_8wave": round(at_8, 2),
                 "frac_of_box_peak_at_kernel_occupancy": round(valu["achieved"] / at_kernel, 4), "frac_of_box_peak": round(valu["achieved"] / at_8, 4),
                 "box_peak_samples": [{"%dwave" % k: {"t_lane_ops_s": v[0], "shader_mhz": v[1], "slowest_xcd_mhz": v[2]} for k, v in sorted(smp.items())} for smp in samples]})
    probe_mhz = mean(waves, 1)
    kernel_mhz = kernel_clock[0] if kernel_clock else None
    if kernel_mhz and probe_mhz:
        # per CLOCK: what a box that merely clocks lower under this kernel's load does not change
        k_per_clk = valu["achieved"] * 1e12 / (kernel_mhz * 1e6) / PEAK_LANE_OPS_PER_CLOCK
        p_per_clk = at_kernel * 1e12 / (probe_mhz * 1e6) / PEAK_LANE_OPS_PER_CLOCK
        valu.update({"shader_mhz_during_timed_launches": round(kernel_mhz, 1), "slowest_xcd_mhz_during_timed_launches": round(kernel_clock[1], 1),
                     "shader_mhz_during_probe": round(probe_mhz, 1), "slowest_xcd_mhz_during_probe": round(mean(waves, 2), 1),
                     "frac_of_issue_peak_per_clock": round(k_per_clk, 4), "probe_frac_of_issue_peak_per_clock": round(p_per_clk, 4),
                     "frac_of_box_peak_per_clock": round(k_per_clk / p_per_clk, 4)})
    return valu


def time_config(torch, np, ssim_amd, synth, ctx, dev, name, w, h, pairs, want_map, mode, kats, steps):
    """One extra BASELINE config on this rank: KAT gate, then `steps` launches timed with HIP events on the launch stream."""
    batch = Batch(torch, ssim_amd, synth, ctx, dev, w, h, 0, pairs, want_map)
    sums = torch.zeros(pairs, dtype=torch.float64, device=dev)
    ctx.set_mode(mode)
    plan = ssim_amd.get_plan(w, h, pairs, ctx)
    try:
        ctx.enqueue_batch(batch.params, pairs, sums.data_ptr())
        ctx.synchronize()
        res = ssim_amd.finalize(sums.cpu().numpy(), w, h)
        gate = "kat"
        if mode in (0, 3):
            for i, k in enumerate(kats[:pairs]):
                if mode == 0 and int(res[i].view(np.uint32)) != k:
                    raise SystemExit("%s: known-answer check failed: pair %d -> 0x%08x, want 0x%08x" % (name, i, int(res[i].view(np.uint32)), k))
        elif mode == 1:      # north_star tolerance vs the FMA reference value
            gate = "sanity gate on THIS synthetic image only: |d| <= 1.5e-6 vs the FMA known answer (the mode's contract and where it ends: DESIGN.md section 2; only modes exact / unfused are guarantees)"
            for i, k in enumerate(kats[:pairs]):
                ref = float(np.array([k], np.uint32).view(np.float32)[0])
                if abs(float(res[i]) - ref) > 1.5e-6:
                    raise SystemExit("%s: fast mode off by %.3g on pair %d" % (name, abs(float(res[i]) - ref), i))
        elif mode == 4:      # the reference's test tolerance vs its double oracle
            gate = "sanity gate on THIS synthetic image only: |d| < 2e-6 vs naive<double> (the reference's TEST tolerance; not a north_star-compliance claim: DESIGN.md section 2)"
            for i, nv in enumerate(NAIVE_KATS.get((w, h), ())[:pairs]):
                if abs(float(res[i]) - nv) >= 2e-6:
                    raise SystemExit("%s: separable mode off by %.3g on pair %d" % (name, abs(float(res[i]) - nv), i))
        else:                # fp64 internals: the naive<double> value, rounded to float
            gate = "|d| <= 6e-8 vs naive<double>"
            if (w, h) == (4096, 4096) and abs(float(res[0]) - NAIVE_F64_4K) > 6e-8 + 1e-9:
                raise SystemExit("%s: double mode off by %.3g" % (name, abs(float(res[0]) - NAIVE_F64_4K)))
        if want_map:
            mm = float(batch.imgs[0][2].double().mean().item())
            if abs(mm - float(res[0])) > 1e-6:
                raise SystemExit("%s: map mean %.9f disagrees with the global SSIM %.9f" % (name, mm, float(res[0])))
        t_settle = time.perf_counter()          # the clock has dropped while the host checked the results: settle, untimed
        while time.perf_counter() - t_settle < 0.15:
            ctx.enqueue_batch(batch.params, pairs, sums.data_ptr())
            ctx.synchronize()
        for _ in range(2):
            ctx.enqueue_batch(batch.params, pairs, sums.data_ptr())
        ctx.synchronize()
        occ = (KERNEL_WAVES_PER_SIMD[mode], 8)
        box = [b for b in [probe_box_or_none(ctx, occ)] if b] if mode != 2 else []
        ctx.get_profile()
        ctx.set_profiling(True)
        ctx.get_profile_clock()          # clears the clock counters
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.enqueue_batch(batch.params, pairs, sums.data_ptr())
        ctx.synchronize()
        wall = time.perf_counter() - t0
        ctx.set_profiling(False)
        n, ms = ctx.get_profile()
        k_clock = ctx.get_profile_clock()[:2]
        if mode != 2:
            box += [b for b in [probe_box_or_none(ctx, occ)] if b]
    finally:
        ctx.set_mode(0)
    k_ms = ms / max(n, 1)
    roof, valu = figures(mode, pairs, w, h, want_map, k_ms)
    against_box(valu, mode, box, k_clock)
    if mode == 2 and k_clock[0]:
        valu["shader_mhz_during_timed_launches"], valu["slowest_xcd_mhz_during_timed_launches"] = round(k_clock[0], 1), round(k_clock[1], 1)
    return {"workload": "%d x %dx%d%s" % (pairs, w, h, " + map" if want_map else ""), "mode": MODE_NAMES[mode], "gate": gate,
            "kernel": kernel_name(mode, 0, want_map, plan), "kernel_avg_ms": round(k_ms, 4), "launches_timed": int(n),
            "mpix_s": round(float(pairs) * w * h / (k_ms * 1e-3) / 1e6, 1),
            "mpix_s_wall": round(float(pairs) * w * h * steps / wall / 1e6, 1),
            "roofline": roof, "valu": valu}


def shard_table(args, world):
    """Which global pairs [first, last) each of `world` ranks owns, and the batch size: the one place the timed run takes
    its sharding from."""
    from ssim_amd import sharding
    _, _, weak_pairs, strong_total, _, _, _ = WORKLOADS[args.workload]
    if args.scaling == "weak":
        per = args.pairs or weak_pairs
        return {"scaling": "weak", "total": world * per, "shards": [list(sharding.shard_range(r, world, per)) for r in range(world)]}
    total = args.pairs or strong_total
    return {"scaling": "strong", "total": total, "shards": [list(x) for x in sharding.split_batch(total, world)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200: ~0.5 s of GPU time on the default workload, long enough for an outside utilisation sampler to see it)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="4k")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU per step (weak) / in total (strong); 0: the workload's default")
    ap.add_argument("--mode", type=int, default=0, help="0 exact (default), 1 fast (reference-order E planes, separable mu), 2 double, 3 unfused, 4 separable")
    ap.add_argument("--strip-rows", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--exchange", choices=["auto", "native", "torch"], default="auto",
                    help="N > 1: who carries the all-reduce of the per-image sums.  native = the product's own rmgr_ssim_hip_comm_* (RCCL behind "
                         "the C ABI; rank 0's communicator id travels over the launcher's process group); torch = torch.distributed.all_reduce; "
                         "auto (default) = native when its communicator comes up on every rank within the deadline, else torch -- the line says which")
    ap.add_argument("--sustain", type=float, default=6.0, metavar="SECONDS",
                    help="after the timed region: the same step repeated for about SECONDS on every rank (the same number of steps on all of them), "
                         "reported as `sustained` -- throughput at thermal / clock steady state, and long enough for an outside utilisation sampler "
                         "to see the GPUs busy (0: off; never part of `value`)")
    ap.add_argument("--watchdog", type=float, default=900.0, metavar="SECONDS", help="dump all Python stacks and exit non-zero if the run takes longer than this (0: off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold-start", action="store_true", help="skip the cold-start / concurrent-caller child processes of single_pair")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs after the headline")
    ap.add_argument("--print-launch", action="store_true", help="print the rank launcher command for --gpus N and exit (no GPU needed)")
    ap.add_argument("--print-shards", action="store_true", help="print the shard table (JSON: which global pairs each rank owns) for --gpus N and exit (no GPU needed)")
    ap.add_argument("--cpu-baseline-only", type=float, default=0.0, metavar="SECONDS",
                    help="internal: time the CPU baseline for about SECONDS and print its JSON object (the child process of the default run)")
    args = ap.parse_args()
    if args.cpu_baseline_only > 0:
        print(json.dumps(cpu_baseline(args.cpu_baseline_only)))
        return 0
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.print_launch:
        print(" ".join(launcher_command(args.gpus, [a for a in sys.argv[1:] if a != "--print-launch"])))
        return 0
    if args.print_shards:
        print(json.dumps(shard_table(args, args.gpus)))
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, sys.argv[1:])

    # A run that stops making progress (a collective whose peer died, a box that lost its GPU) must end with a diagnosis, not
    # sit until somebody's outer limit kills it silently: after --watchdog seconds every thread's Python stack goes to stderr
    # and the process exits non-zero.  (The native exchange has its own, much shorter deadlines.)
    if args.watchdog > 0:
        import faulthandler
        faulthandler.enable()
        faulthandler.dump_traceback_later(args.watchdog, exit=True)

    # stdout carries exactly one line, the JSON of rank 0: native libraries write banners to file descriptor 1 (RCCL
    # prints its version block there), so everything else is sent to stderr and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import ssim_amd
    from ssim_amd import sharding, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not os.path.exists(ssim_amd.LIB_PATH) and local_rank == 0:
        subprocess.run(["make", "-C", ROOT, "lib"], check=True, stdout=subprocess.DEVNULL)   # built artefacts normally travel with the tree
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher started a different number of ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X: no HIP device visible (there is no CPU fallback)")
    # TEST MODE (never set by a launcher): SSIM_BENCH_SHARED_DEVICE=1 puts every rank on device 0 and carries the control plane and the
    # exchange over gloo (RCCL refuses two ranks on one GPU) -- so that a 1-GPU box can run the N > 1 code path end to end (shards,
    # per-rank diagnosis, digest comparison, the exchange, max-over-ranks timing).  The line it prints says so and is not a result.
    shared_device = os.environ.get("SSIM_BENCH_SHARED_DEVICE") == "1"
    dev_index = 0 if shared_device else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit("rank %d has no device: %d visible" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or os.environ.get("SSIM_BENCH_FORCE_DIST") == "1":      # the env switch lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29511))
        if shared_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    # Everything (torch ops, RCCL, our launches) is ordered on one explicit non-default stream: the legacy
    # NULL stream adds implicit synchronisation to every launch.
    if os.environ.get("SSIM_BENCH_NULL_STREAM") != "1":
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    stream = torch.cuda.current_stream()
    ctx = ssim_amd.Context(dev_index, ctypes.c_void_p(stream.cuda_stream), mode=args.mode)
    ctx.set_tuning(args.strip_rows, args.variant)

    # --- the exchange step's carrier.  The product's own is rmgr_ssim_hip_comm_*: rank 0 creates the
    #     communicator id, the launcher's process group -- the control plane: barriers, the max over ranks of the elapsed
    #     time -- carries its 128 bytes to the other ranks, and every rank joins with a DEADLINE (a rank that cannot
    #     come up returns ETIMEDOUT instead of hanging the job; then, under `auto`, all ranks agree to use torch's). ---
    exchange = {"carrier": "none", "ranks_seen": world if dist is None else None}
    if dist is not None:
        exchange = {"carrier": "torch", "requested": args.exchange, "ranks_seen": dist.get_world_size()}
        if args.exchange in ("auto", "native"):
            os.environ.setdefault("RMGR_SSIM_HIP_COMM_TIMEOUT_S", "60")
            err = None
            try:
                uid = sharding.handoff_unique_id(dist, ssim_amd.Context.comm_unique_id, rank)
                ctx.comm_init(uid, world, rank)
            except Exception as e:       # noqa: BLE001 -- any failure of the native carrier is reported, never fatal under auto
                err = "%s: %s" % (type(e).__name__, e)
            if sharding.all_agree(dist, err is None, dev):
                exchange.update({"carrier": "native", "ranks_seen": ctx.comm_rank_count(), "rccl": ssim_amd.Context.comm_describe()})
                if exchange["ranks_seen"] != world:
                    raise SystemExit("rank %d: RCCL counts %d ranks in the native communicator, the launcher started %d" % (rank, exchange["ranks_seen"], world))
            else:
                if err is None:
                    ctx.comm_destroy()
                exchange["native_error"] = err or "another rank's communicator did not come up"
                sys.stderr.write("bench.py rank %d: native exchange unavailable (%s)\n" % (rank, exchange["native_error"]))
                if args.exchange == "native":
                    raise SystemExit("--exchange native: %s" % exchange["native_error"])
    native = exchange["carrier"] == "native"

    W, H, weak_pairs, strong_total, want_map, kats, workload_desc = WORKLOADS[args.workload]
    # --- shard the batch by image: rank r owns global pairs [first, last) ---
    table = shard_table(args, world)
    total = table["total"]
    first, last = table["shards"][rank]
    mine = last - first
    batch = Batch(torch, ssim_amd, synth, ctx, dev, W, H, first, mine, want_map)
    headline_plan = ssim_amd.get_plan(W, H, mine, ctx) if mine else None
    sums_all = torch.zeros(total, dtype=torch.float64, device=dev)       # zero except this rank's slice
    work = torch.zeros_like(sums_all)
    my_slice_ptr = sums_all.data_ptr() + 8 * first

    ex_events = []                       # (begin, end) HIP events around the exchange of each timed step

    def step(timed=False, carrier=None):
        if mine:
            ctx.enqueue_batch(batch.params, mine, my_slice_ptr)
        if dist is None:
            return sums_all
        # all-reduce of the per-image partial sums; other ranks contribute exact zeros
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        use_native = native if carrier is None else carrier == "native"
        out = sharding.exchange_sums_native(ctx, sums_all, work) if use_native else sharding.exchange_sums(sums_all, work, dist)
        if timed:
            e1.record()
            ex_events.append((e0, e1))
        return out

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # --- known-answer gate: the reference's FMA-path float bits for the first seeds, on whichever rank owns them,
    #     and (after the exchange) on every rank for the whole vector's plausibility ---
    full = step()
    fence()
    full_bits = full.cpu().numpy().view(np.uint64).copy()
    res = ssim_amd.finalize(full.cpu().numpy(), W, H)
    if args.mode == 0:
        for i, k in enumerate(kats):
            if i < total and int(res[i].view(np.uint32)) != k:
                raise SystemExit("known-answer check failed on rank %d: pair %d -> %r (0x%08x), want 0x%08x"
                                 % (rank, i, float(res[i]), int(res[i].view(np.uint32)), k))
    if not np.all(np.isfinite(res)) or res.min() < 0.85 or res.max() > 0.95:
        raise SystemExit("implausible batch results on rank %d: %r" % (rank, res))
    if want_map and mine:   # the map that was written must average to the global value
        mm = float(batch.imgs[0][2].double().mean().item())
        if abs(mm - float(res[first])) > 1e-6:
            raise SystemExit("map mean %.9f disagrees with the global SSIM" % mm)
    result_digest = "%016x" % (int(np.bitwise_xor.reduce(res.view(np.uint32).astype(np.uint64) * (np.arange(res.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1)))) & 0xFFFFFFFFFFFFFFFF)
    # --- self-diagnosis of a multi-rank run, BEFORE anything is timed: every rank says which device it bound (index + PCI bus id), which
    #     carrier the exchange uses and how many ranks RCCL counts, and the digest of the result vector it holds after the exchange.  The
    #     all-reduce gives every rank every sum: a rank whose digest differs from rank 0's computed something else, and the run aborts
    #     non-zero instead of printing a number (the first N > 1 run on hardware is the driver's: it must explain itself) ---
    try:
        props = torch.cuda.get_device_properties(dev)
        pci = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))
    except Exception:  # noqa: BLE001
        pci = "unknown"
    who = "device %d (pci %s, %s) pairs [%d, %d) carrier %s ranks_seen %s" % (dev_index, pci, torch.cuda.get_device_name(dev), first, last, exchange["carrier"], exchange["ranks_seen"])
    digests_ok, rank_lines = sharding.compare_digests(dist, result_digest, who)
    if rank == 0 and (world > 1 or not digests_ok):
        for l in rank_lines:
            sys.stderr.write("bench.py: %s\n" % l)
        sys.stderr.flush()
    if not digests_ok:
        raise SystemExit("bench.py: the ranks hold different result vectors after the exchange (see the per-rank lines above): no number is reported")
    exchange["per_rank"] = rank_lines

    # Clock settle (untimed, before the W warm-up steps): the chip takes tens of milliseconds of sustained load
    # to leave its idle DVFS state; a ~1 ms step measured right after the uploads reads up to 20 % slow.
    # Every step carries a collective when there are several ranks, so the ranks must leave this loop after the SAME number of
    # steps: each looks at its own clock, and they stop when all of them have seen the quarter second (one MIN all-reduce of a flag
    # per four steps) -- a rank that stopped on its own clock one step before its peers would leave them in an all-reduce nobody answers.
    # The box's own VALU peak (rmgr_ssim_hip_probe_valu), the yardstick the kernel's lane-operations per second are divided by: probed BEFORE the clock-settle
    # loop and again right after the timed steps.  Not between the settle loop and the timed steps: a pure packed-FMA stream at full occupancy draws more power than
    # the SSIM kernel, and the steps that follow it must be the kernel's own steady state, not the probe's aftermath ($SSIM_BENCH_PROBE=late puts it there for the A/B,
    # off skips it).  Local to the rank, no collective.
    occupancies = (2, 3, 8)
    probe_when = os.environ.get("SSIM_BENCH_PROBE", "early")
    box_samples = [b for b in [probe_box_or_none(ctx, occupancies)] if b] if probe_when == "early" else []
    t_settle = time.perf_counter()
    while True:
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        settled = time.perf_counter() - t_settle >= 0.25
        if dist is not None:
            settled = sharding.all_agree(dist, settled, dev)
        if settled:
            break
    if probe_when == "late":
        box_samples += [b for b in [probe_box_or_none(ctx, occupancies)] if b]
    for _ in range(args.warmup):
        step()
    fence()
    # HIP events bracket the main kernel of every timed step, on the stream it is launched on
    # (rmgr_ssim_hip_set_profiling): two event records per ~3 ms step, read back after the fence.
    ctx.get_profile()
    ctx.set_profiling(True)
    ctx.get_profile_clock()              # allocates / clears the clock counters (untimed)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    fence()
    elapsed = time.perf_counter() - t0
    ctx.set_profiling(False)
    # the last step must have delivered the vector the gate checked, bit for bit (full_bits: a host SNAPSHOT taken
    # at the gate -- `full` itself is the tensor the timed steps keep writing)
    last = work if dist is not None else sums_all
    if not np.array_equal(last.cpu().numpy().view(np.uint64), full_bits):
        raise SystemExit("rank %d: the timed steps returned different sums than the gated step" % rank)
    launches, kernel_ms = ctx.get_profile()
    kernel_avg_ms = kernel_ms / max(launches, 1)
    kernel_clock = ctx.get_profile_clock()[:2]
    if probe_when != "off":
        box_samples += [b for b in [probe_box_or_none(ctx, occupancies)] if b]
    if dist is not None:
        # every rank's own clock and kernel time, for the record (the number below is the MAX over ranks: one slow GPU sets it, and this says which)
        mine_ms = [None] * world
        dist.all_gather_object(mine_ms, (rank, round(elapsed / args.steps * 1e3, 4), round(kernel_avg_ms, 4)))
        mine_ms.sort()
        exchange["per_rank_ms"] = [{"rank": r, "ms_per_step": e, "kernel_avg_ms": k} for r, e, k in mine_ms]
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the exchange as the stream saw it: copy of the vector + the all-reduce, INCLUDING the wait for the slowest rank
        ex_ms = sorted(a.elapsed_time(b) for a, b in ex_events)
        exchange.update({"allreduce_ms_per_step": round(sum(ex_ms) / max(len(ex_ms), 1), 4), "allreduce_ms_median": round(ex_ms[len(ex_ms) // 2], 4) if ex_ms else None,
                         "allreduce_note": "HIP events around the exchange (vector copy + all-reduce of %d doubles) of every timed step on rank 0: includes waiting for the slowest rank's kernels" % total})
        # the carrier that was NOT timed must deliver the same vector, bit for bit (untimed)
        if native:
            chk = step(carrier="torch")
            fence()
            if not np.array_equal(chk.cpu().numpy().view(np.uint64), full_bits):
                raise SystemExit("rank %d: the torch exchange returned different sums than the native one" % rank)
            exchange["crosscheck"] = "torch.distributed.all_reduce delivers the same vector bit for bit (untimed step)"

    # --- sustained rate: the same step (kernel + exchange) for about --sustain seconds, the same number of steps on every rank
    #     (derived from the all-reduced elapsed time, so the collectives pair up); never `value` ---
    sustained = {}
    if args.sustain > 0:
        n_sus = max(args.steps, int(args.sustain / max(elapsed / args.steps, 1e-6)))
        fence()
        power = PowerSampler() if rank == 0 else None      # rocm-smi in a side thread (a child process per sample): what the package draws and clocks under this very load
        if power:
            power.start()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            step()
        fence()
        dt_sus = time.perf_counter() - t0
        if power:
            power.finish()
        if dist is not None:
            t = torch.tensor([dt_sus], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_sus = float(t.item())
        sustained = {"steps": n_sus, "seconds": round(dt_sus, 3), "ms_per_step": round(dt_sus / n_sus * 1e3, 4),
                     "mpix_s": round(float(total) * W * H * n_sus / dt_sus / 1e6, 1),
                     "vs_value": round((elapsed / args.steps) / (dt_sus / n_sus), 4),
                     "note": "the timed step repeated back to back for ~%.0f s (max over ranks): throughput at clock / thermal steady state; not `value`" % args.sustain}
        if power and power.summary():
            sustained["package"] = power.summary()
        last = work if dist is not None else sums_all
        if not np.array_equal(last.cpu().numpy().view(np.uint64), full_bits):
            raise SystemExit("rank %d: the sustained steps returned different sums than the gated step" % rank)

    # --- the two opt-in modes on the same batch (kernel time only; never `value`) ---
    other = {}
    if args.mode == 0 and rank == 0 and mine:
        for key, m in (("fast_mode", 1), ("separable_mode", 4)):
            ctx.set_mode(m)
            plan_m = ssim_amd.get_plan(W, H, mine, ctx)
            for _ in range(2):
                ctx.enqueue_batch(batch.params, mine, my_slice_ptr)
            ctx.synchronize()
            ctx.set_profiling(True)
            for _ in range(min(max(args.steps // 2, 3), 100)):
                ctx.enqueue_batch(batch.params, mine, my_slice_ptr)
            ctx.synchronize()
            n_f, ms_f = ctx.get_profile()
            clock_f = ctx.get_profile_clock()[:2]
            ctx.set_profiling(False)
            ctx.set_mode(0)
            roof_f, valu_f = figures(m, mine, W, H, want_map, ms_f / n_f)
            if box_samples:
                against_box(valu_f, m, box_samples, clock_f)
            other[key] = {"mode": MODE_NAMES[m], "kernel": kernel_name(m, args.variant, want_map, plan_m),
                          "kernel_avg_ms": round(ms_f / n_f, 4), "mpix_s": round(float(mine) * W * H / (ms_f / n_f * 1e-3) / 1e6, 1),
                          "roofline_frac": roof_f["frac"], "valu_frac": valu_f["frac"], "ops_per_pixel": valu_f["ops_per_pixel"],
                          "valu_frac_of_box_peak_at_kernel_occupancy": valu_f.get("frac_of_box_peak_at_kernel_occupancy"), "kernel_waves_per_simd": valu_f.get("kernel_waves_per_simd"),
                          "shader_mhz": valu_f.get("shader_mhz_during_timed_launches"), "valu_frac_of_box_peak_per_clock": valu_f.get("frac_of_box_peak_per_clock")}
        ctx.enqueue_batch(batch.params, mine, my_slice_ptr)
        ctx.synchronize()

    # --- single-pair latency/throughput (BASELINE.json configs[1] literally: one pair per call) ---
    single = {}
    if rank == 0 and mine:
        imgs = batch.imgs
        p0 = ssim_amd.make_params(W, H, imgs[0][0].data_ptr(), 1, W, imgs[0][1].data_ptr(), 1, W)   # global-only
        one = (ssim_amd.Params * 1)(p0)
        for _ in range(5):
            ctx.enqueue_batch(one, 1, my_slice_ptr)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(50):
            ctx.enqueue_batch(one, 1, my_slice_ptr)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 50
        t1 = time.perf_counter()
        for _ in range(20):
            ctx.compute_device(p0)
        dts = (time.perf_counter() - t1) / 20
        # the unchanged reference call: HOST pointers, pageable memory, PCIe staging included
        ha, hb = imgs[0][0].cpu().numpy(), imgs[0][1].cpu().numpy()
        ha_keep, hb_keep = ha, hb
        hv, _ = ssim_amd.compute_ssim(ha, hb)
        if args.mode == 0:
            assert int(hv.view(np.uint32)) == kats[0]
        def best_and_median(fn, n):
            fn()
            ts = []
            for _ in range(n):
                t_ = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t_)
            return min(ts), statistics.median(ts)
        dth, dth_med = best_and_median(lambda: ssim_amd.compute_ssim(ha, hb), 6)
        # ... and with the per-pixel map copied back into a caller-owned pageable buffer (banded pipeline)
        hmap = np.zeros((H, W), np.float32)
        hv2, _ = ssim_amd.compute_ssim(ha, hb, out_map=hmap)
        assert int(hv2.view(np.uint32)) == int(hv.view(np.uint32)) and abs(float(hmap.mean(dtype=np.float64)) - float(hv)) < 1e-6
        dthm, dthm_med = best_and_median(lambda: ssim_amd.compute_ssim(ha, hb, out_map=hmap), 6)
        # a batch of host-resident pairs through the pipelined entry point (PCIe staging overlapped with the kernels)
        nb = max(2, min(8, (256 << 20) // (2 * W * H)))
        hp = [(ha, hb)] * nb
        ssim_amd.compute_ssim_batch(hp)
        t1 = time.perf_counter()
        hbv = ssim_amd.compute_ssim_batch(hp)
        dtb = (time.perf_counter() - t1) / nb
        assert int(hbv[0].view(np.uint32)) == int(hv.view(np.uint32))
        single = {"enqueued_ms": round(dt * 1e3, 4), "enqueued_mpix_s": round(W * H / dt / 1e6, 1),
                  "host_batch_pairs": nb, "host_batch_ms_per_pair": round(dtb * 1e3, 3), "host_batch_mpix_s": round(W * H / dtb / 1e6, 1),
                  "blocking_call_ms": round(dts * 1e3, 4), "blocking_call_mpix_s": round(W * H / dts / 1e6, 1),
                  "host_pointer_call_ms": round(dth * 1e3, 3), "host_pointer_call_mpix_s": round(W * H / dth / 1e6, 1),
                  "host_pointer_call_median_ms": round(dth_med * 1e3, 3),
                  "host_pointer_call_with_map_ms": round(dthm * 1e3, 3), "host_pointer_call_with_map_mpix_s": round(W * H / dthm / 1e6, 1),
                  "host_pointer_call_with_map_median_ms": round(dthm_med * 1e3, 3),
                  "host_pointer_note": "unchanged rmgr_ssim_compute_ssim on pageable host memory, PCIe staging included; best and median of 6 calls"}
        del hmap, ha, hb
        # what a one-shot caller pays (the reference's shipped callers make 1-4 calls per process), and what concurrent callers get:
        # fresh CHILD processes (never an exec of this one), torch-free (tools/cold_start_probe.py, tools/concurrent_callers.py)
        if not args.no_cold_start and world == 1:
            def child(*cmd):
                try:
                    r = subprocess.run([sys.executable] + list(cmd), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120)
                    return json.loads(r.stdout.decode().strip().splitlines()[-1]) if r.returncode == 0 else {"failed": r.returncode}
                except Exception as e:  # noqa: BLE001 -- a probe must not cost the bench its line
                    return {"failed": repr(e)}
            probe = os.path.join(ROOT, "tools", "cold_start_probe.py")
            plain = [child(probe, "plain") for _ in range(2)]
            best = min((p for p in plain if "total_ms" in p), key=lambda p: p["total_ms"], default={})
            single["cold_first_call_ms"] = best.get("total_ms")
            single["cold_start"] = {
                "what": "fresh process: dlopen(librmgr-ssim-hip.so) -> first rmgr_ssim_compute_ssim() on a 1080p pair in pageable memory -> return; best of 2 processes",
                "plain": best, "split": child(probe, "split"), "cli_bbb1080_png_vs_jpeg50": child(probe, "cli"),
                "note": "split: runtime_init_ms = hipInit + device enumeration, context_ms = first hipSetDevice + stream (the HIP runtime's queue), "
                        "code_object_ms = first launch of any kernel of the library = the load of its whole code object (all strip-kernel instantiations), "
                        "first_ssim_ms vs steady_ssim_ms = first launch of the strip kernel + reduction"}
            single["concurrent_callers"] = child(os.path.join(ROOT, "tools", "concurrent_callers.py"), "4096", "1", "1,2,4", "1.0", "--json")
            single["concurrent_callers_1080p"] = child(os.path.join(ROOT, "tools", "concurrent_callers.py"), "1920", "1", "1,2,4", "1.0", "--json")
            single["concurrent_callers_note"] = ("N host threads looping the unchanged rmgr_ssim_compute_ssim WITH the map on pageable memory, each call on a leased default "
                                                 "context; 4096^2 + map moves 33.5 MB in and 67 MB out per call: at the ~56 GB/s a pageable copy attains on this link one "
                                                 "direction alone caps the call rate near 14 k Mpix/s, so one thread already sits at ~75 % of it")
        # --- one process, every visible device (rmgr_ssim_hip_compute_ssim_batch_host_devices: a worker thread + context per device, the host batch sharded
        #     by image, no exchange step in one address space).  Silent on a one-GPU box; on the driver's 8-GPU node its N = 1 line then carries BOTH multi-GPU
        #     forms: this leg and, in the N = 2, 4, 8 lines, one process per GPU + RCCL.  Host pointers: PCIe staging included, never `value`.
        #     $SSIM_BENCH_FORCE_DEVICES="0,0" lets a one-GPU box exercise the leg (several contexts on one device). ---
        forced = os.environ.get("SSIM_BENCH_FORCE_DEVICES")
        devs = [int(x) for x in forced.split(",")] if forced else list(range(torch.cuda.device_count()))
        if world == 1 and len(devs) >= 2:
            try:
                per_dev = max(2, min(8, (256 << 20) // (2 * W * H)))
                hp = [(ha_keep, hb_keep)] * (per_dev * len(devs))
                one = ssim_amd.compute_ssim_batch_devices(hp[:per_dev], devs[:1], args.mode)
                t1 = time.perf_counter()
                one = ssim_amd.compute_ssim_batch_devices(hp[:per_dev], devs[:1], args.mode)
                dt_one = time.perf_counter() - t1
                allv = ssim_amd.compute_ssim_batch_devices(hp, devs, args.mode)
                t1 = time.perf_counter()
                allv = ssim_amd.compute_ssim_batch_devices(hp, devs, args.mode)
                dt_all = time.perf_counter() - t1
                if not (np.all(allv.view(np.uint32) == one.view(np.uint32)[0]) and (args.mode != 0 or int(one.view(np.uint32)[0]) == kats[0])):
                    raise SystemExit("single_process_devices: the devices disagree: %r" % [hex(int(x)) for x in allv.view(np.uint32)])
                single["single_process_devices"] = {
                    "devices": devs, "forced": bool(forced), "pairs_per_device": per_dev, "pairs": len(hp),
                    "one_device_mpix_s": round(per_dev * W * H / dt_one / 1e6, 1), "all_devices_mpix_s": round(len(hp) * W * H / dt_all / 1e6, 1),
                    "speedup_vs_one_device": round((len(hp) / dt_all) / (per_dev / dt_one), 3),
                    "note": "rmgr_ssim_hip_compute_ssim_batch_host_devices from ONE process: host-resident pairs (pageable memory, PCIe staging pipelined per device) sharded by image over "
                            "the devices, one worker thread + context each; every result bit-identical to the one-device call; second of two calls timed"}
            except ssim_amd.SsimError as e:        # a device that cannot be opened, a failed allocation ...: the leg is reported as failed, the line survives (a DISAGREEMENT between the devices, above, is fatal)
                single["single_process_devices"] = {"devices": devs, "failed": str(e)}
        ctx.enqueue_batch(batch.params, mine, my_slice_ptr)      # restore the slice for consistency
        torch.cuda.synchronize()

    # --- the other BASELINE configs, rank 0 only, after the timed region (kernel time from HIP events) ---
    configs = {}
    if rank == 0 and world == 1 and not args.no_configs and args.mode == 0 and args.variant == 0 and args.strip_rows == 0:
        del batch                                  # free the headline batch first
        torch.cuda.empty_cache()
        ksteps = min(max(args.steps // 2, 5), 100)
        w8, h8 = 8192, 8192
        configs["8k-map exact"] = time_config(torch, np, ssim_amd, synth, ctx, dev, "8k-map exact", w8, h8, 2, True, 0, WORKLOADS["8k-map"][5], ksteps)
        configs["8k-map fast"] = time_config(torch, np, ssim_amd, synth, ctx, dev, "8k-map fast", w8, h8, 2, True, 1, WORKLOADS["8k-map"][5], ksteps)
        configs["8k-map separable"] = time_config(torch, np, ssim_amd, synth, ctx, dev, "8k-map separable", w8, h8, 2, True, 4, WORKLOADS["8k-map"][5], ksteps)
        configs["1080p x128 exact"] = time_config(torch, np, ssim_amd, synth, ctx, dev, "1080p exact", 1920, 1080, 128, False, 0, WORKLOADS["1080p"][5], ksteps)
        configs["1080p x128 fast"] = time_config(torch, np, ssim_amd, synth, ctx, dev, "1080p fast", 1920, 1080, 128, False, 1, WORKLOADS["1080p"][5], ksteps)
        configs["1080p x128 separable"] = time_config(torch, np, ssim_amd, synth, ctx, dev, "1080p separable", 1920, 1080, 128, False, 4, WORKLOADS["1080p"][5], ksteps)
        configs["4k double + map"] = time_config(torch, np, ssim_amd, synth, ctx, dev, "4k double", 4096, 4096, 4, True, 2, WORKLOADS["4k"][5], ksteps)
        configs["4k x1 exact"] = time_config(torch, np, ssim_amd, synth, ctx, dev, "4k single", 4096, 4096, 1, False, 0, WORKLOADS["4k"][5], 50)

    # --- plan_regret: is the untuned default plan the best of its candidates on THIS box?  rmgr_ssim_hip_tune on a context of its own (the timed
    #     context above ran, and stays on, the untuned default): kernel time of the default plan / of the best candidate, candidates interleaved ---
    plan_regret = {}
    if rank == 0 and world == 1 and not args.no_configs and args.variant == 0 and args.strip_rows == 0:
        with ssim_amd.Context(dev_index, ctypes.c_void_p(stream.cuda_stream), mode=args.mode) as tctx:
            try:
                for key, (tw, th, tn, tmap) in (("headline", (W, H, mine, want_map)), ("1080p x128", (1920, 1080, 128, False)), ("8k-map x2", (8192, 8192, 2, True)), ("4k x1", (4096, 4096, 1, False))):
                    if not tn:
                        continue
                    r = tctx.tune(tw, th, tn, tmap)
                    plan_regret[key] = {"workload": "%d x %dx%d%s" % (tn, tw, th, " + map" if tmap else ""), "default_ms": round(r["default_ms"], 4), "best_ms": round(r["best_ms"], 4),
                                        "regret": round(r["default_ms"] / r["best_ms"], 4), "best": "default" if r["best"] == (0, 0) else "variant %d, strip rows %d" % r["best"],
                                        "candidates": [{"variant": v, "strip_rows": rr, "ms": round(ms, 4)} for v, rr, ms in r["candidates"]]}
            except ssim_amd.SsimError as e:          # an allocation that did not fit, a failed launch: reported, the line survives
                plan_regret["failed"] = str(e)
        plan_regret["note"] = ("rmgr_ssim_hip_tune in this process: the candidate plans plan() chooses between (candidates[0] = the untuned default the timed steps ran), interleaved over three "
                               "rounds on synthetic pairs of the shape; regret = default / best kernel time (1.0: the default is the best; a candidate replaces it only beyond 0.5 %)")

    attainable = None
    if rank == 0:
        attainable = attainable_hbm_gbs(torch, dev)
    if rank == 0:
        pixels = float(total) * W * H * args.steps
        value = pixels / elapsed / 1e6
        roof, valu = figures(args.mode, mine, W, H, want_map, kernel_avg_ms) if mine else ({}, {})
        if roof:
            traffic, traffic_note = measured_traffic(args.mode, args.workload, mine, ssim_amd.kernel_source_id())
            roof.update({"traffic": traffic, "traffic_note": traffic_note,
                         "attainable_copy": attainable, "attainable_note": "device-to-device copy of 1 GiB on this box (read + write bytes / time), GB/s",
                         "kernel": kernel_name(args.mode, args.variant, want_map, headline_plan), "kernel_avg_ms": round(kernel_avg_ms, 4), "launches_timed": int(launches),
                         "note": "HBM fraction reported because the metric asks for it; what binds this kernel is not HBM: per shader cycle it issues ~95 % of what a pure packed-FMA stream issues at its "
                                 "occupancy (valu.frac_of_box_peak_per_clock), and its clock is set by package power (sustained.package: ~1360 W of the 1400 W cap under this load)"})
            if box_samples:
                against_box(valu, args.mode, box_samples, kernel_clock)
            valu.update({"box_peak_2wave": max(b[2][0] for b in box_samples) if box_samples else None,
                         "box_peak_note": "rmgr_ssim_hip_probe_valu in this process, on this box: a pure v_pk_fma_f32 stream at a FORCED occupancy of 2 / 8 waves per SIMD "
                                          "(register footprint padded, grid = the chip's capacity; 40 ms of untimed launches, then the median of 5 launches of ~2 ms each), once before the "
                                          "clock-settle loop that precedes the warm-up steps and once right after the timed steps (box_peak_samples, in that order); the fractions divide by the BETTER of the two (a probe burst can run in a degraded mode, never in an enhanced one: rmgr/ssim-hip.h); "
                                          "shader_mhz_* / slowest_xcd_mhz_*: the clock the timed strip-kernel launches / the probe's timed launches really ran at, mean over the XCDs and the slowest XCD's "
                                          "(one workgroup per XCD counts s_memtime cycles per s_memrealtime tick: rmgr_ssim_hip_get_profile_clock); frac_of_issue_peak_per_clock = lane-operations per shader cycle over the 32768 the chip can issue; "
                                          "frac_of_box_peak_per_clock = that figure over the probe's: what a box that only clocks lower under this kernel's load leaves unchanged",
                         "round4_box_constants": dict(ROUND4_BOX_VALU_TOPS, note="what rounds 4-5 divided by (one round-4 box, tools/occupancy_probe.hip); for comparison only")})
        line = {
            "metric": "Mpix/s (global SSIM, no map) on 4K pairs; achieved HBM GB/s vs roofline",
            "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64" if args.mode == 2 else "f32", "data": "synthetic",
            "config": {"workload": "%s, %s, sharded by image, %s"
                                   % (workload_desc,
                                      "%d pairs per GPU per step" % mine if args.scaling == "weak" else "%d pairs in total split over %d GPUs (%d on rank 0)" % (total, world, mine),
                                      ("RCCL all-reduce of per-image fp64 sums per step (%s)" % ("rmgr_ssim_hip_comm_allreduce_sums behind the C ABI" if native else "torch.distributed"))
                                      if dist is not None else "single GPU, no collective"),
                       "name": args.workload, "mode": MODE_NAMES[args.mode],
                       "pairs_per_gpu": mine, "pairs_total": total, "width": W, "height": H, "strip_rows": args.strip_rows, "variant": args.variant,
                       "results_digest": result_digest},
            "exchange": exchange,
            "roofline": roof,
            "valu": valu,
            "sustained": sustained,
            "single_pair": single,
            "fast_mode": other.get("fast_mode", {}),
            "separable_mode": other.get("separable_mode", {}),
            "configs": configs,
            "plan_regret": plan_regret,
            "device": ctx.describe(),
        }
        if shared_device:
            line["test_mode"] = ("SSIM_BENCH_SHARED_DEVICE=1: all %d ranks ran on ONE GPU with gloo as the carrier -- a functional run of the N > 1 code path; "
                                 "`value` is NOT a scaling result" % world)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_in_child()
            # the CPU figure of each BASELINE config beside its GPU line (all host threads, best run; the same image, the same arithmetic contract)
            per = line["cpu_baseline"].get("per_config", {})
            beside = {"8k-map exact": ("8k-map", "threads_all_mpix_s"), "1080p x128 exact": ("1080p-batch", "threads_all_mpix_s"),
                      "4k double + map": ("4k-double", "with_map_threads_all_mpix_s"), "4k x1 exact": ("4k", "threads_all_mpix_s")}
            for name, (key, field) in beside.items():
                if name in configs and key in per and field in per[key]:
                    configs[name]["cpu_mpix_s"] = per[key][field]
                    configs[name]["cpu_source"] = "cpu_baseline.per_config['%s'].%s" % (key, field)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
