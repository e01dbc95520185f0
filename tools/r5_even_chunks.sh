#!/bin/bash
# Round 5: launches whose balanced chunk divides the strip column evenly (or is a whole number of columns): chunks (tuning variant 6) = one round of taller strips without
# a seam.  Against the shipped strips (0) and both row-sum forms (2 = in the blur phase, 3 = EARLY), interleaved per shape.
cd "$(dirname "$0")/.."
for M in 0 1; do
for P in 8 16 32 64 128; do python3 tools/ab.py $P 4096 $M 0 0,2,3,6 5 0 | tail -4 | sed "s/^/mode $M $P x 4096: /"; done
for P in 2 4 8 16; do python3 tools/ab.py $P 8192 $M 0 0,2,3,6 5 0 | tail -4 | sed "s/^/mode $M $P x 8192: /"; done
for P in 32 128 512; do python3 tools/ab.py $P 2048 $M 0 0,2,3,6 5 0 | tail -4 | sed "s/^/mode $M $P x 2048: /"; done
done
for P in 256 1024; do python3 tools/ab.py $P 1920 0 0 0,2,3 5 0 1080 | tail -3 | sed "s/^/mode 0 $P x 1080p: /"; done
for P in 1024 4096; do python3 tools/ab.py $P 512 0 0 0,2,3 5 0 | tail -3 | sed "s/^/mode 0 $P x 512: /"; done
