#!/bin/bash
# Round 5: plan()'s cap on the strip height of the two-waves-per-SIMD kernels (512 rows since round 2) against 1024 and 2048, default tuning, builds interleaved per shape.
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-r5_strip_cap}; mkdir -p $OUT
{
  for M in 0 1; do
    for P in 4 8 12 16 24 32 48 64 128; do tools/ab_libs.sh "C512 C1024 C2048" $P 4096 $M 0 0 2; done
    for P in 1 2 4 8 16; do tools/ab_libs.sh "C512 C1024 C2048" $P 8192 $M 0 0 2; done
  done
  for P in 2 4 8; do tools/ab_libs.sh "C512 C1024 C2048" $P 8192 0 1 0 2; done
  for P in 64 256 1024; do tools/ab_libs.sh "C512 C1024 C2048" $P 1920 0 0 0 2 1080; done
  for P in 8 32; do tools/ab_libs.sh "C512 C1024 C2048" $P 2048 0 0 0 2; done
} > $OUT/sweep.txt 2>&1
