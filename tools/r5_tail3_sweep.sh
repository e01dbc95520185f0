#!/bin/bash
# Round 5: plan()'s occupancy factors of the three-waves-per-SIMD kernels (tail3) re-fitted to the round-5 separable kernel: the strips plan() picks with the old
# factors (libT0: 490, 715) against the new (libT1: 450, 665; libT2: 430, 640), MODE_SEPARABLE, default tuning, interleaved per shape.   usage: tools/r5_tail3_sweep.sh <out>
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-r5_tail3}; mkdir -p $OUT; LIBS=${2:-"T0 T1 T2"}
{
  for P in 1 2 3 4 6 8 12 16 24 32 64; do tools/ab_libs.sh "$LIBS" $P 4096 4 0 0 2; done
  for P in 1 2 4 8; do tools/ab_libs.sh "$LIBS" $P 8192 4 0 0 2; done
  for P in 1 2 4; do tools/ab_libs.sh "$LIBS" $P 8192 4 1 0 2; done
  for P in 1 4 8 16 24 32 48 64 96 128 192 256; do tools/ab_libs.sh "$LIBS" $P 1920 4 0 0 2 1080; done
  for P in 16 64 256 1024; do tools/ab_libs.sh "$LIBS" $P 512 4 0 0 2; done
  for P in 4 8 16; do tools/ab_libs.sh "$LIBS" $P 4096 2 1 0 2; done
} > $OUT/sweep.txt 2>&1
