#!/usr/bin/env python3
"""Is the SSIM kernel power-bound?  Runs (a) the headline batch (32 x 4096^2, MODE_EXACT) and (b) the forced-occupancy FMA probe back to back for a few seconds each while a
thread samples `rocm-smi` (socket power, shader clock, temperature, the power cap), and prints what the chip drew and clocked under each load, next to the clock the
kernels themselves measured (rmgr_ssim_hip_get_profile_clock / rmgr_ssim_hip_probe_valu).     usage: python3 tools/power_probe.py [seconds=3]"""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402
from ssim_amd import synth  # noqa: E402


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
        card = d[sorted(d)[0]]
        return {k: v for k, v in card.items() if any(s in k.lower() for s in ("power", "sclk", "temperature (sensor junction)", "temperature (sensor edge)", "temperature (sensor memory)"))}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


class Sampler(threading.Thread):
    def __init__(self):
        threading.Thread.__init__(self)
        self.samples, self.stop = [], False

    def run(self):
        while not self.stop:
            self.samples.append(smi())
            time.sleep(0.05)


def summarise(name, samples):
    keys = sorted({k for s in samples for k in s})
    print("%s: %d rocm-smi samples" % (name, len(samples)))
    for k in keys:
        vals = []
        for s in samples:
            v = str(s.get(k, "")).strip("()").replace("Mhz", "").replace("MHz", "")
            try:
                vals.append(float(v))
            except ValueError:
                pass
        if vals:
            vals.sort()
            print("    %-55s min %.1f  median %.1f  max %.1f" % (k, vals[0], vals[len(vals) // 2], vals[-1]))
        else:
            print("    %-55s %s" % (k, samples[-1].get(k)))


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    print("idle:", smi())
    ctx = ssim_amd.Context(0)
    w = h = 4096
    n = 32
    params = (ssim_amd.Params * n)()
    keep = []
    for i in range(n):
        da, db = ctx.alloc(w * h), ctx.alloc(w * h)
        ctx.synth_pair(da.ptr, w, db.ptr, w, w, h, synth.BASE_SEED + i)
        keep += [da, db]
        params[i] = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w)
    sums = ctx.alloc(8 * n)
    for mode, variant, label in ((0, 0, "SSIM kernel, MODE_EXACT, 32 x 4096^2"), (4, 0, "SSIM kernel, MODE_SEPARABLE, 32 x 4096^2"), (1, 0, "SSIM kernel, MODE_FAST, 32 x 4096^2"),
                                 (2, 0, "SSIM kernel, MODE_DOUBLE (one column per lane), 32 x 4096^2"), (0, 1, "SSIM kernel, MODE_EXACT, one column per lane (variant 1), 32 x 4096^2"),
                                 (0, 3, "SSIM kernel, MODE_EXACT, strips with EARLY row sums (variant 3), 32 x 4096^2")):
        ctx.set_mode(mode)
        ctx.set_tuning(0, variant)
        s = Sampler()
        ctx.set_profiling(True)
        ctx.get_profile_clock()
        t0 = time.perf_counter()
        s.start()
        while time.perf_counter() - t0 < secs:
            for _ in range(20):
                ctx.enqueue_batch(params, n, sums.ptr)
            ctx.synchronize()
        s.stop = True
        s.join()
        launches, ms = ctx.get_profile()
        mhz, lo, _ = ctx.get_profile_clock()
        ctx.set_profiling(False)
        print("%s: %.4f ms per launch = %.1f Gpix/s; the kernel measured %.0f MHz (slowest XCD %.0f)" % (label, ms / launches, n * w * h / (ms / launches) / 1e6, mhz, lo))
        pw = sorted(float(x.get("Current Socket Graphics Package Power (W)", 0)) for x in s.samples[4:])
        if pw and pw[len(pw) // 2] > 0:
            print("    energy: %.2f nJ per pixel at the median %.0f W" % (pw[len(pw) // 2] * (ms / launches) * 1e-3 / (n * w * h) * 1e9, pw[len(pw) // 2]))
        summarise("  " + label, s.samples[4:])
    ctx.set_mode(0)
    ctx.set_tuning(0, 0)
    for waves in (2, 3, 8):
        s = Sampler()
        t0 = time.perf_counter()
        s.start()
        rates = []
        while time.perf_counter() - t0 < secs:
            rates.append(ctx.probe_valu(waves, 0, 20, with_clock=True))
        s.stop = True
        s.join()
        best = max(rates)
        print("FMA probe at %d waves per SIMD: %.2f T lane-ops/s at %.0f MHz (slowest XCD %.0f); %d calls" % (waves, best[0], best[1], best[2], len(rates)))
        summarise("  probe %d waves" % waves, s.samples[4:])
    ctx.close()


if __name__ == "__main__":
    main()
