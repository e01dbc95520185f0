#!/bin/bash
# Round 5: cache policy of the map stores (SSIM_MAP_STORE_AUX: 0 = default, 1 = sc0, 2 = nt (shipped), 3 = sc0 nt, 18 = sc1 nt), 2 x 8192^2 + map
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-r5_mapstore}; mkdir -p $OUT
{
  echo "# separable, with map";  tools/ab_libs.sh "aux0 aux1 aux2 aux3 aux18" 2 8192 4 1 0 2
  echo "# separable, no map";    tools/ab_libs.sh "aux2" 2 8192 4 0 0 2
  echo "# exact, with map";      tools/ab_libs.sh "aux0 aux2 aux3 aux18" 2 8192 0 1 0 2
  echo "# exact, no map";        tools/ab_libs.sh "aux2" 2 8192 0 0 0 2
} > $OUT/ab.txt 2>&1
