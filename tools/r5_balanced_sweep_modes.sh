#!/bin/bash
# Round 5: the balanced schedule (tuning variant 6) in MODE_FAST (1) and MODE_SEPARABLE (4) against their strips (variant 0), no map,
# interleaved in one process per launch shape (tools/ab.py).   usage: tools/r5_balanced_sweep_modes.sh <out-subdir>
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-r5_balanced_modes}; mkdir -p $OUT
{
  python3 tools/balanced_check.py
  for M in 4 1; do
    for P in 8 16 24 32 40 48 64 80 96 128 160 192 256 384; do python3 tools/ab.py $P 1920 $M 0 0,6 5 0 1080; done
    for P in 1 2 3 4 6 8 12 16 24 32 64; do python3 tools/ab.py $P 4096 $M 0 0,6 5 0; done
    for P in 1 2 4; do python3 tools/ab.py $P 8192 $M 0 0,6 3 0; done
    for P in 64 256 1024; do python3 tools/ab.py $P 512 $M 0 0,6 3 0; done
  done
} > $OUT/sweep.txt 2>&1
