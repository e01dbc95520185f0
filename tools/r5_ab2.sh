#!/bin/bash
# Round 5: round-4 library (build/ab/libr4.so) against the working tree's (libnew.so): odd left halo (6 window reads per plane),
# MODE_SEPARABLE's epilogue / conversion.  usage: tools/r5_ab2.sh <out-subdir>
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-r5_ab2}; mkdir -p $OUT
{
  echo "# 32 x 4096^2, separable";        tools/ab_libs.sh "r4 new" 32 4096 4 0 0 3
  echo "# 2 x 8192^2 + map, separable";   tools/ab_libs.sh "r4 new" 2 8192 4 1 0 3
  echo "# 128 x 1080p, separable";        tools/ab_libs.sh "r4 new" 128 1920 4 0 0 3 1080
  echo "# 32 x 4096^2, exact";            tools/ab_libs.sh "r4 new" 32 4096 0 0 0 3
  echo "# 2 x 8192^2 + map, exact";       tools/ab_libs.sh "r4 new" 2 8192 0 1 0 2
  echo "# 128 x 1080p, exact";            tools/ab_libs.sh "r4 new" 128 1920 0 0 0 3 1080
  echo "# 1 x 4096^2, exact";             tools/ab_libs.sh "r4 new" 1 4096 0 0 0 2
  echo "# 32 x 4096^2, hybrid";           tools/ab_libs.sh "r4 new" 32 4096 1 0 0 2
  echo "# 32 x 4096^2, unfused";          tools/ab_libs.sh "r4 new" 32 4096 3 0 0 1
} > $OUT/ab.txt 2>&1
