#!/usr/bin/env python3
"""plan()'s untuned default against the measured best candidate (rmgr_ssim_hip_tune) over the launch shapes round 5 fitted its rules on
(tools/sweep_variants.sh: ten sizes x sixteen batch sizes) -- run on a box that was NOT used for the fit.  Prints one line per shape and a summary:
on how many shapes the default is within 1 % of the best candidate, and the regret (default / best kernel time) where it is not.

usage: python3 tools/tune_sweep.py [mode=0] [with_map=0] [quick | big]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402

SIZES = [(1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (1600, 1200), (5120, 2880), (1000, 1000), (3000, 2000), (7680, 4320), (640, 480), (4096, 4096), (8192, 8192)]
COUNTS = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256]


def main():
    mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    with_map = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
    quick = "quick" in sys.argv
    big = "big" in sys.argv          # only the long launches the video sweep leaves out (> 1.1 Gpixel)
    ctx = ssim_amd.Context(0, mode=mode)
    print("# %s; mode %d, map %d" % (ctx.describe(), mode, with_map))
    rows = []
    shapes = [(w, h, n) for (w, h) in SIZES for n in (COUNTS[::3] if quick else COUNTS) if 4000000 <= w * h * n <= 1100000000]
    if big:
        shapes = [(1920, 1080, 512), (1920, 1080, 1024), (4096, 4096, 96), (4096, 4096, 128), (2048, 2048, 512), (8192, 8192, 24), (8192, 8192, 32), (1280, 720, 1024), (3840, 2160, 192), (512, 512, 4096)]
    for (w, h, n) in shapes:
        if True:
            px = w * h * n
            r = ctx.tune(w, h, n, with_map)
            p = ssim_amd.get_plan(w, h, n, ctx)
            regret = r["default_ms"] / r["best_ms"]
            rows.append((w, h, n, regret, r))
            print("%5d x %4dx%-4d default %8.4f ms (%6.1f Gpix/s; %s)  best %8.4f ms %-28s regret %.4f   candidates: %s"
                  % (n, w, h, r["default_ms"], px / r["default_ms"] / 1e6, "chunks %d x %d rows, interleave %d" % (p.balancedChunks, p.balancedChunkRows, p.balancedInterleave) if p.balancedChunks and not with_map else "strips of %d rows" % p.stripRows,
                     r["best_ms"], "(the default)" if r["best"] == (0, 0) else "variant %d rows %d" % r["best"], regret,
                     " ".join("%d/%d:%.4f" % c for c in r["candidates"])))
            sys.stdout.flush()
            ctx.clear_tuned()
    within = sum(1 for x in rows if x[3] <= 1.01)
    worst = sorted(rows, key=lambda x: -x[3])[:8]
    print("# %d shapes; default within 1 %% of the best candidate on %d (%.1f %%); mean regret %.4f; worst: %s"
          % (len(rows), within, 100.0 * within / max(len(rows), 1), sum(x[3] for x in rows) / max(len(rows), 1),
             ", ".join("%d x %dx%d %.3f" % (x[2], x[0], x[1], x[3]) for x in worst)))
    ctx.close()


if __name__ == "__main__":
    main()
