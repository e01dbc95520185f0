#!/bin/bash
# Round 5: MODE_SEPARABLE, round-4 library against the working tree's (tools/ab_libs.sh: build/ab/libr4.so, libnew.so), and
# inside the new library the pair-staging kernel (variant 0) against the byte-load two-column kernel (variant 2).
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-r5_sep_ab}; mkdir -p $OUT
{
  echo "# 32 x 4096^2, separable";        tools/ab_libs.sh "r4 new" 32 4096 4 0 0 3
  echo "# 2 x 8192^2 + map, separable";   tools/ab_libs.sh "r4 new" 2 8192 4 1 0 3
  echo "# 128 x 1080p, separable";        tools/ab_libs.sh "r4 new" 128 1920 4 0 0 3 1080
  echo "# 1 x 4096^2, separable";         tools/ab_libs.sh "r4 new" 1 4096 4 0 0 2
  echo "# new library, variants 0 (pair staging) / 2 (byte loads) / 1 (one column), interleaved"
  RMGR_SSIM_LIB=$PWD/build/ab/libnew.so python3 tools/ab.py 32 4096 4 0 0,2,1 5 0
  RMGR_SSIM_LIB=$PWD/build/ab/libnew.so python3 tools/ab.py 2 8192 4 0 0,2 5 1
  RMGR_SSIM_LIB=$PWD/build/ab/libnew.so python3 tools/ab.py 128 1920 4 0 0,2 5 0 1080
  echo "# exact / hybrid / double unchanged?"
  tools/ab_libs.sh "r4 new" 32 4096 0 0 0 2
  tools/ab_libs.sh "r4 new" 32 4096 1 0 0 2
} > $OUT/ab.txt 2>&1
