#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: Mpix/s of compute_ssim (global SSIM,
no map) on 4096x4096 uint8 pairs, with the achieved-vs-roofline figures and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: PAIRS_PER_GPU distinct synthetic 4096x4096
pairs per GPU (BASELINE.json configs[1] image; seeds 0x5EED+i), resident in HBM,
one batched launch through the C ABI (rmgr_ssim_hip_enqueue_batch), and -- for N > 1 -- one RCCL
all-reduce of the per-image fp64 sums so that every rank holds every result (images are sharded
by rank, weak scaling: per-GPU work is fixed).  Timing: barrier + synchronize on both sides of
exactly K steps, max over ranks; value = all pixels of all ranks / that time.

Before any timing the results are gated on the reference's known answer for pair 0
(FMA path: 0x3f64b7be = 0.893428683).

Only the cpu_baseline leg touches oracle/: it times the real reference kernels (oracle/_ref,
kind "reference") or, where that prebuilt library is absent, the C restatement (kind "port").
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W = H = 4096
PAIRS_PER_GPU = 32
KAT_PAIR0_HEX = 0x3f64b7be            # reference FMA path on seed 0x5EED (tests/golden/manifest.json)
BYTES_PER_PIXEL = 2                   # algorithmic HBM bytes, global-only: one uint8 from each image
# BASELINE.json configs as selectable workloads; the default ("4k") is the one the metric is quoted on.
#   name: (width, height, pairs per GPU, write map, KAT of pair 0 = reference FMA float bits, description)
WORKLOADS = {
    "4k":     (4096, 4096, 32, False, 0x3f64b7be, "4096x4096 uint8 pairs (BASELINE.json configs[1] image), global SSIM only"),
    "8k-map": (8192, 8192, 2, True, 0x3f64b5b4, "8192x8192 uint8 pairs with per-pixel SSIM map writeback (BASELINE.json configs[2])"),
    "1080p":  (1920, 1080, 128, False, 0x3f64bb1f, "1920x1080 uint8 pairs, global SSIM only (BASELINE.json configs[3]: 1024 pairs over 8 GPUs = 128 per GPU)"),
}
HBM_PEAK_GBS = 8000.0                 # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# fp32 VALU work of MODE_EXACT per output pixel: 5 planes x (5 fold adds + 6 mul + 30 fma
# + 10 ring adds) + 23 for the SSIM formula/divide/fp64 accumulate = 278 lane-ops
VALU_OPS_PER_PIXEL = 278
VALU_PEAK_TOPS = 78.6                 # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz lane-ops/s; = 157.3 TFLOP/s fp32 vector spec / 2
VALU_MEASURED_PEAK_TOPS = 68.7        # best v_pk_fma_f32 rate tools/valu_probe.hip reaches on this chip: 8 waves/SIMD (profiles/r01_valu_probe.txt)
VALU_MEASURED_2WAVE_TOPS = 58.1       # the same probe at the 2 waves/SIMD the kernel's 110 accumulator VGPRs allow


def cpu_baseline(budget_s=12.0):
    """Reference FMA+OpenMP path (or the port) on this box's host cores, one 4096^2 pair."""
    import numpy as np
    import oracle
    a, b = oracle.synth_pair(W, H, 0x5EED)
    if oracle.have_ref():
        kind = "reference"
        cores = min(oracle.ref_lib().ref_max_threads(), 64)      # the reference caps its pool at 64 (src/ssim.cpp:1025)
        fn = lambda: oracle.ref_ssim(a, b, impl=5, threads=cores)
    else:
        kind = "port"
        cores = oracle.oracle_lib().oracle_max_threads()
        fn = lambda: oracle.ssim_f32(a, b, fused=True, threads=cores)
    v = fn()[0]
    assert int(np.float32(v).view(np.uint32)) == KAT_PAIR0_HEX, "CPU baseline disagrees with the known answer"
    times = []
    t_stop = time.perf_counter() + budget_s          # a bounded sample: ~12 s of all-core work
    while time.perf_counter() < t_stop:
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    t1 = time.perf_counter()
    oracle_1t = oracle.ref_ssim(a, b, impl=5, threads=1) if kind == "reference" else oracle.ssim_f32(a, b, threads=1)
    t_single = time.perf_counter() - t1
    del oracle_1t
    best = min(times)
    model, procs = host_cpu()
    return {"value": round(W * H / best / 1e6, 1), "unit": "Mpix/s", "cores": cores, "kind": kind,
            "cpu_model": model, "logical_cpus": procs, "one_thread_mpix_s": round(W * H / t_single / 1e6, 1),
            "sample": "%d back-to-back runs of one 4096x4096 pair (seed 0x5EED) over ~12 s, best run reported; median %.1f Mpix/s; 1 thread %.1f Mpix/s; %s"
                      % (len(times), W * H / statistics.median(times) / 1e6, W * H / t_single / 1e6,
                         "real reference FMA/AVX kernel objects (oracle/_ref) driven by the harness tile loop, OpenMP static schedule"
                         if kind == "reference" else "oracle/ssim_oracle.c restatement, OpenMP")}


def measured_traffic(mode, workload, pairs):
    """HBM bytes per launch from the committed PMC measurement (profiles/traffic.json), scaled to
    this batch; None when no measurement exists for the configuration."""
    key = {"4k": "exact_4096_nomap", "8k-map": "exact_8192_map", "1080p": "exact_1080p_nomap"}.get(workload)
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        if mode == 0 and key in t:
            return float(t[key]["bytes_per_pair"]) * pairs
    except (OSError, ValueError, KeyError):
        pass
    return None


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="4k")
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU per step (0: the workload's default)")
    ap.add_argument("--mode", type=int, default=0, help="0 exact (default), 1 fast separable")
    ap.add_argument("--strip-rows", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import ssim_amd
    from ssim_amd import sharding, synth

    if not os.path.exists(ssim_amd.LIB_PATH) and int(os.environ.get("LOCAL_RANK", "0")) == 0:
        import subprocess                       # built artefacts normally travel with the tree; rebuild if they did not
        subprocess.run(["make", "-C", ROOT, "lib"], check=True, stdout=subprocess.DEVNULL)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("SSIM_BENCH_FORCE_DIST") == "1":      # the env switch lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    # Everything (torch ops, RCCL, our launches) is ordered on one explicit non-default stream: the legacy
    # NULL stream adds implicit synchronisation to every launch.
    if os.environ.get("SSIM_BENCH_NULL_STREAM") != "1":
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    stream = torch.cuda.current_stream()
    ctx = ssim_amd.Context(local_rank, ctypes.c_void_p(stream.cuda_stream), mode=args.mode)
    ctx.set_tuning(args.strip_rows, args.variant)

    global W, H, KAT_PAIR0_HEX, BYTES_PER_PIXEL
    W, H, default_pairs, want_map, KAT_PAIR0_HEX, workload_desc = WORKLOADS[args.workload]
    BYTES_PER_PIXEL = 6 if want_map else 2       # + one float per pixel when the map is written
    P = args.pairs or default_pairs
    # --- resident synthetic batch: rank r owns global pairs r*P .. r*P+P-1 ---
    first, _ = sharding.shard_range(rank, world, P)
    imgs = []
    params = (ssim_amd.Params * P)()
    for i in range(P):
        a, b = synth.pair_torch(W, H, synth.BASE_SEED + first + i, device=dev)
        m = torch.empty((H, W), dtype=torch.float32, device=dev) if want_map else None
        imgs.append((a, b, m))
        params[i] = ssim_amd.make_params(W, H, a.data_ptr(), 1, W, b.data_ptr(), 1, W, m.data_ptr() if want_map else None, 1, W)
    sums_all = torch.zeros(world * P, dtype=torch.float64, device=dev)       # zero except this rank's slice
    work = torch.zeros_like(sums_all)
    my_slice_ptr = sums_all.data_ptr() + 8 * first

    def step():
        ctx.enqueue_batch(params, P, my_slice_ptr)
        # all-reduce of the per-image partial sums; other ranks contribute exact zeros
        return sharding.exchange_sums(sums_all, work, dist)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # --- known-answer gate ---
    full = step()
    fence()
    res = ssim_amd.finalize(full.cpu().numpy(), W, H)
    if int(res[0].view(np.uint32)) != KAT_PAIR0_HEX:
        raise SystemExit("known-answer check failed: pair 0 -> %r (0x%08x), want 0x%08x" % (float(res[0]), int(res[0].view(np.uint32)), KAT_PAIR0_HEX))
    if not np.all(np.isfinite(res)) or res.min() < 0.85 or res.max() > 0.95:
        raise SystemExit("implausible batch results: %r" % res)
    if want_map:   # the map that was written must average to the global value
        mm = float(imgs[0][2].double().mean().item())
        if abs(mm - float(res[0 if rank == 0 else first])) > 1e-6:
            raise SystemExit("map mean %.9f disagrees with the global SSIM" % mm)

    # Clock settle (untimed, before the W warm-up steps): the chip takes tens of milliseconds of sustained load
    # to leave its idle DVFS state; a ~1 ms step measured right after the uploads reads up to 20 % slow.
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.25:
        step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    # HIP events bracket the main kernel of every timed step, on the stream it is launched on
    # (rmgr_ssim_hip_set_profiling): two event records per ~3 ms step, read back after the fence.
    ctx.get_profile()
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ctx.set_profiling(False)
    launches, kernel_ms = ctx.get_profile()
    kernel_avg_ms = kernel_ms / max(launches, 1)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # --- the opt-in separable mode on the same batch (kernel time only; never `value`) ---
    other = {}
    if args.mode == 0:
        ctx.set_mode(1)
        for _ in range(2):
            ctx.enqueue_batch(params, P, my_slice_ptr)
        ctx.synchronize()
        ctx.set_profiling(True)
        for _ in range(max(args.steps // 2, 3)):
            ctx.enqueue_batch(params, P, my_slice_ptr)
        ctx.synchronize()
        n_f, ms_f = ctx.get_profile()
        ctx.set_profiling(False)
        ctx.set_mode(0)
        other = {"mode": "fast (separable fp32, within tolerance, not bit-identical)", "kernel_avg_ms": round(ms_f / n_f, 4),
                 "mpix_s": round(float(P) * W * H / (ms_f / n_f * 1e-3) / 1e6, 1)}
        ctx.enqueue_batch(params, P, my_slice_ptr)
        ctx.synchronize()

    # --- single-pair latency/throughput (BASELINE.json configs[1] literally: one pair per call) ---
    single = {}
    if rank == 0:
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
        hv, _ = ssim_amd.compute_ssim(ha, hb)
        assert int(hv.view(np.uint32)) == KAT_PAIR0_HEX
        t1 = time.perf_counter()
        for _ in range(5):
            ssim_amd.compute_ssim(ha, hb)
        dth = (time.perf_counter() - t1) / 5
        # a batch of host-resident pairs through the pipelined entry point (PCIe staging overlapped with the kernels)
        nb = max(2, min(8, (256 << 20) // (2 * W * H)))
        hp = [(ha, hb)] * nb
        ssim_amd.compute_ssim_batch(hp)
        t1 = time.perf_counter()
        hbv = ssim_amd.compute_ssim_batch(hp)
        dtb = (time.perf_counter() - t1) / nb
        assert int(hbv[0].view(np.uint32)) == KAT_PAIR0_HEX
        single = {"enqueued_ms": round(dt * 1e3, 4), "enqueued_mpix_s": round(W * H / dt / 1e6, 1),
                  "host_batch_pairs": nb, "host_batch_ms_per_pair": round(dtb * 1e3, 3), "host_batch_mpix_s": round(W * H / dtb / 1e6, 1),
                  "blocking_call_ms": round(dts * 1e3, 4), "blocking_call_mpix_s": round(W * H / dts / 1e6, 1),
                  "host_pointer_call_ms": round(dth * 1e3, 3), "host_pointer_call_mpix_s": round(W * H / dth / 1e6, 1)}
        ctx.enqueue_batch(params, P, my_slice_ptr)      # restore the slice for consistency
        torch.cuda.synchronize()

    attainable = None
    if rank == 0:
        attainable = attainable_hbm_gbs(torch, dev)
    if rank == 0:
        pixels = float(world) * P * W * H * args.steps
        value = pixels / elapsed / 1e6
        bytes_per_launch = float(P) * W * H * BYTES_PER_PIXEL
        achieved = bytes_per_launch / (kernel_avg_ms * 1e-3) / 1e9
        ops_px = VALU_OPS_PER_PIXEL if args.mode in (0, 3) else 133      # separable: 5 x 22 blur + 23
        valu = ops_px * float(P) * W * H / (kernel_avg_ms * 1e-3) / 1e12
        line = {
            "metric": "Mpix/s (global SSIM, no map) on 4K pairs; achieved HBM GB/s vs roofline",
            "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %d pairs per GPU per step, sharded by image, %s"
                                   % (workload_desc, P, "RCCL all-reduce of per-image fp64 sums per step" if world > 1 else "single GPU, no collective"),
                       "name": args.workload,
                       "mode": ["exact (reference FMA order, bit-faithful)", "fast (separable fp32)", "double", "unfused"][args.mode],
                       "pairs_per_gpu": P, "width": W, "height": H, "strip_rows": args.strip_rows, "variant": args.variant},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(args.mode, args.workload, P),
                         "attainable_copy": attainable, "attainable_note": "device-to-device copy of 1 GiB on this box (read + write bytes / time), GB/s",
                         "kernel": "ssim_strip1_kernel" if (args.mode == 2 or args.variant == 1) else "ssim_strip2_kernel", "kernel_avg_ms": round(kernel_avg_ms, 4), "launches_timed": int(launches),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "kernel is fp32-VALU bound (see valu); HBM fraction reported because the metric asks for it"},
            "valu": {"achieved": round(valu, 2), "peak": VALU_PEAK_TOPS, "unit": "T lane-ops/s", "frac": round(valu / VALU_PEAK_TOPS, 4),
                     "ops_per_pixel": ops_px, "measured_peak": VALU_MEASURED_PEAK_TOPS,
                     "frac_of_measured_peak": round(valu / VALU_MEASURED_PEAK_TOPS, 4),
                     "measured_peak_at_kernel_occupancy": VALU_MEASURED_2WAVE_TOPS,
                     "frac_of_peak_at_kernel_occupancy": round(valu / VALU_MEASURED_2WAVE_TOPS, 4)},
            "single_pair": single,
            "fast_mode": other,
            "device": ctx.describe(),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
