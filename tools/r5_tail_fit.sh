#!/bin/bash
# Round 5: how fast does a wavefront run when its SIMD holds 1, 2 or 3 of them?  Launches of exactly 1024 / 2048 / 3072 strips of 512 rows (4 / 8 / 12 x 4096^2,
# strip height forced): the time of such a launch IS the time of one strip at that occupancy.  What plan()'s packing model calls tail2[] / tail3[].
cd "$(dirname "$0")/.."
for M in 4 0 1; do for P in 4 8 12 16 20 24; do python3 tools/ab.py $P 4096 $M 512 0 5 0 | tail -1 | sed "s/^/mode $M pairs $P rows 512: /; s/variant 0: //; s/ssim.*//"; done; done
