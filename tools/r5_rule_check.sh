#!/bin/bash
# the balanced rule at 3.5 % (libR35, shipped) against 7 % (libR70) on the shapes that change hands, default tuning, builds interleaved per shape (exact, then hybrid)
cd "$(dirname "$0")/.."
for M in 0 1; do
for S in "16 1920 1080" "32 1920 1080" "80 1920 1080" "12 1920 1080" "8 3840 2160" "12 3840 2160" "16 3840 2160" "12 2560 1440" "24 2560 1440" "32 2560 1440" "48 2560 1440" "64 2560 1440" \
         "16 1600 1200" "128 1600 1200" "16 5120 2880" "6 4096 4096" "16 3000 2000" "32 3000 2000" "24 1280 720" "128 1280 720" "192 1280 720" "3 7680 4320" "192 1000 1000" "256 640 480" "24 1000 1000"; do
  set -- $S
  tools/ab_libs.sh "R70 R35" $1 $2 $M 0 0 2 $3
done; done
