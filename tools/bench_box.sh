#!/bin/bash
# Runs ON THE GPU BOX: `python bench.py` alone (the driver's command plus the default step count), for the box spread of tools/pick_median_box.py.
# usage: tools/bench_box.sh <out-subdir-of-gpurun_out>
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-box}; mkdir -p "$OUT"
timeout 900 python3 bench.py > "$OUT/bench_4k.log" 2>&1
grep '"metric"' "$OUT/bench_4k.log" > "$OUT/bench_4k.json"
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cold-start --no-configs > "$OUT/bench_driver_flags.log" 2>&1
grep '"metric"' "$OUT/bench_driver_flags.log" > "$OUT/bench_driver_flags.json"
python3 -c "
import json
for f in ('bench_4k.json', 'bench_driver_flags.json'):
    d = json.loads(open('$OUT/' + f).read().strip().splitlines()[-1]); v = d['valu']
    print(f, d['value'], d['roofline']['kernel_avg_ms'], v.get('box_peak_2wave'), v.get('frac_of_box_peak_at_kernel_occupancy'), v.get('shader_mhz_during_timed_launches'), v.get('frac_of_box_peak_per_clock'))
"
