#!/bin/bash
# Round 5: MODE_SEPARABLE held at TWO waves per SIMD (8 KiB of unused LDS per wavefront; plan() packing 2048 wave slots) against three (libW0), default tuning.
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-r5_two_waves}; mkdir -p $OUT
{
  for P in 1 2 4 8 12 16 24 32 64; do tools/ab_libs.sh "W0 W2" $P 4096 4 0 0 2; done
  for P in 1 2 4 8; do tools/ab_libs.sh "W0 W2" $P 8192 4 0 0 2; done
  for P in 1 2 4; do tools/ab_libs.sh "W0 W2" $P 8192 4 1 0 2; done
  for P in 1 8 16 32 64 128 256 1024; do tools/ab_libs.sh "W0 W2" $P 1920 4 0 0 2 1080; done
  for P in 16 64 256 1024; do tools/ab_libs.sh "W0 W2" $P 512 4 0 0 2; done
  tools/ab_libs.sh "W0 W2" 8 1920 4 1 0 2 1080; tools/ab_libs.sh "W0 W2" 4 4096 4 1 0 2
} > $OUT/sweep.txt 2>&1
