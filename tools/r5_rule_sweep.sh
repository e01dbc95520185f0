#!/bin/bash
# Round 5: the balanced schedule's rule on sizes it was not calibrated on (720p ... 5K, odd sizes): strips with row sums in the blur phase (2) / EARLY (3) against chunks (6) and the default (0),
# MODE_EXACT, no map, interleaved per shape.   usage: tools/r5_rule_sweep.sh <out-subdir>
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-r5_rule}; mkdir -p $OUT
{
for S in "1280 720" "1920 1080" "2560 1440" "3840 2160" "1600 1200" "5120 2880" "1000 1000" "3000 2000" "7680 4320" "640 480"; do
  set -- $S
  for P in 1 2 3 4 6 8 12 16 24 32 48 64 96 128 192 256; do
    if [ $(( $1 * $2 * P )) -le 1100000000 ] && [ $(( $1 * $2 * P )) -ge 4000000 ]; then python3 tools/ab.py $P $1 0 0 0,2,3,6 3 0 $2; fi
  done
done
} > $OUT/sweep.txt 2>&1
