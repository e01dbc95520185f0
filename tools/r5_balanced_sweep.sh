#!/bin/bash
# Round 5: where does the balanced schedule (tuning variant 6) beat the strips (2 = row sums in the blur phase, 3 = EARLY row sums)?
# MODE_EXACT, no map, interleaved in one process per launch shape (tools/ab.py).   usage: tools/r5_balanced_sweep.sh <out-subdir>
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-r5_balanced}; mkdir -p $OUT
{
  python3 tools/balanced_check.py
  for P in 8 16 24 32 40 48 64 80 96 128 160 192 256 384; do python3 tools/ab.py $P 1920 0 0 2,3,6 5 0 1080; done
  for P in 1 2 3 4 6 8 12 16 24 32 64; do python3 tools/ab.py $P 4096 0 0 2,3,6 5 0; done
  for P in 1 2 4; do python3 tools/ab.py $P 8192 0 0 2,3,6 3 0; done
  for P in 16 64 256 1024; do python3 tools/ab.py $P 512 0 0 2,3,6 3 0; done
} > $OUT/sweep.txt 2>&1
