#!/bin/bash
# A/B of builds of the library on one box: alternates tools/ab.py runs between build/ab/lib<X>.so files.
# usage: tools/ab_libs.sh "A B" [pairs] [size] [mode] [map] [rows] [reps] [height]
cd "$(dirname "$0")/.."
LIBS=${1:-"A B"}; P=${2:-32}; S=${3:-4096}; M=${4:-0}; MAP=${5:-0}; ROWS=${6:-0}; REPS=${7:-3}; HGT=${8:-$S}
for rep in $(seq $REPS); do
  for L in $LIBS; do
    RMGR_SSIM_LIB=$PWD/build/ab/lib$L.so timeout 300 python3 tools/ab.py $P $S $M $ROWS 0 5 $MAP $HGT 2>&1 | tail -1 | sed "s/^/lib$L p$P s${S}x$HGT m$M map$MAP rows$ROWS: /"
  done
done
