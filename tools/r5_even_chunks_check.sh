#!/bin/bash
# the shipped plan() with the even-chunks rule (libE) against the plan() before it (libN1024), default tuning, interleaved per shape
cd "$(dirname "$0")/.."
for M in 0 1; do
  for P in 1 2 3 4 8 16 32 64; do tools/ab_libs.sh "N1024 E" $P 4096 $M 0 0 2; done
  for P in 1 2 4 8 16; do tools/ab_libs.sh "N1024 E" $P 8192 $M 0 0 2; done
done
for P in 16 64 256; do tools/ab_libs.sh "N1024 E" $P 512 0 0 0 2; done
for P in 64 256 1024; do tools/ab_libs.sh "N1024 E" $P 256 0 0 0 2; done
for P in 8 32 128; do tools/ab_libs.sh "N1024 E" $P 2048 0 0 0 2; done
for P in 4 16 64; do tools/ab_libs.sh "N1024 E" $P 1024 0 0 0 2; done
