#!/bin/bash
# Runs ON THE GPU BOX: does bench.py's in-process VALU probe disturb the timed steps?  bench.py --steps 20 --warmup 5 with $SSIM_BENCH_PROBE = off / early (before the clock-settle loop:
# as shipped) / late (between the settle loop and the warm-up steps), three rounds interleaved on one box (profiles/r06_probe_bimodal.txt, section 5: it does not).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6_probe_ab
for rep in 1 2 3; do for w in off early late; do
  SSIM_BENCH_PROBE=$w timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cold-start --no-cpu-baseline --sustain 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['valu']
print('$w rep $rep value %.1f kernel %.4f ms  2wave %s  frac %s' % (d['value'], d['roofline']['kernel_avg_ms'], v.get('box_peak_2wave'), v.get('frac_of_box_peak_at_kernel_occupancy')))"
done; done > gpurun_out/r6_probe_ab/ab.txt 2>&1
cat gpurun_out/r6_probe_ab/ab.txt
