#!/bin/bash
# ONE parameterised sweep (runs ON THE GPU BOX) in place of round 5's eighteen tools/r5_*.sh: for every launch shape of a named SET and every
# arithmetic mode, either the tuning VARIANTS of the library in the tree interleaved in one process (tools/ab.py), or several BUILDS of the library
# alternated on the box (tools/ab_libs.sh over build/ab/lib<NAME>.so, made with tools/build_ab.sh), or a list of strip HEIGHTS.
#
# usage: tools/sweep_variants.sh <out-subdir-of-gpurun_out> <set> <modes, comma separated> <what> [rounds=3] [map=0]
#   set    1080p | 4k | 8k | 512 | 2048 | video | rule | headline | sep-rows | <a file with "pairs width height" lines>
#   what   variants:0,2,3,6      tuning variants interleaved per shape           (ab.py)
#          libs:"A B C"          builds build/ab/libA.so ... alternated per shape (ab_libs.sh)
#          rows:0,96,184,512     strip heights, default variant                   (ab.py, one run per height)
# The round-5 records and the invocation that reproduces each (profiles/r05_*.txt):
#   r05_balanced_sweep.txt        sweep_variants.sh X 1080p 0 variants:2,3,6 5 ; X 4k 0 variants:2,3,6 5 ; X 8k 0 variants:2,3,6 ; X 512 0 variants:2,3,6
#   r05_balanced_sweep_modes.txt  the same sets with modes 4,1 and variants:0,6
#   r05_rule_sweep.txt            sweep_variants.sh X video 0 variants:0,2,3,6          (fit: tools/r5 rule fit, now tools/tune_sweep.py measures the regret directly)
#   r05_strip_cap_sweep.txt       sweep_variants.sh X 4k 0,1 libs:"C512 C1024 C2048" 2 ; X 8k 0,1 libs:... ; last section: X 4k 0,1 variants:0,2,3,6 5 (+ 8k, 2048)
#   r05_tail3_sweep.txt           sweep_variants.sh X 4k 4 libs:"$LIBS" 2 (+ 8k, 1080p, 512) ; tail fit: X headline 4,0,1 rows:512 5
#   r05_kernel_ab.txt             sweep_variants.sh X headline 4,0,1,3 libs:"r4 new"
#   r05_prefetch_and_map_store_ab.txt   sweep_variants.sh X 8k 4,0 libs:"aux0 aux1 aux2 aux3 aux18" 2 1
#   r05_exact_lds_ablation.txt    sweep_variants.sh X headline 4 libs:"L0 L1 L2" 2        (builds: tools/r5_exact_lds_ablation.patch)
#   r05_two_waves: W0 W2 / W0 W3  sweep_variants.sh X 4k 4 libs:"W0 W2" 2 (+ 8k, 1080p, 512)
#   r05_tall_strips / sep rows    sweep_variants.sh X headline 0,4 rows:0,512,1024,2048,4096 5 ; X sep-rows 4 rows:0,96,136,184,256,320,384,512
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:?out dir}; SET=${2:?set}; MODES=${3:-0}; WHAT=${4:-variants:0}; ROUNDS=${5:-3}; MAP=${6:-0}
mkdir -p "$OUT"
shapes() {
  case "$SET" in
    1080p)    for P in 8 16 24 32 40 48 64 80 96 128 160 192 256 384; do echo "$P 1920 1080"; done ;;
    4k)       for P in 1 2 3 4 6 8 12 16 24 32 48 64 128; do echo "$P 4096 4096"; done ;;
    8k)       for P in 1 2 4 8 16; do echo "$P 8192 8192"; done ;;
    512)      for P in 16 64 256 1024; do echo "$P 512 512"; done ;;
    2048)     for P in 8 32 128 512; do echo "$P 2048 2048"; done ;;
    headline) echo "32 4096 4096"; echo "2 8192 8192"; echo "128 1920 1080"; echo "1 4096 4096" ;;
    rule)     printf '%s\n' "16 1920 1080" "32 1920 1080" "80 1920 1080" "12 1920 1080" "8 3840 2160" "12 3840 2160" "16 3840 2160" "12 2560 1440" "24 2560 1440" "32 2560 1440" \
                "48 2560 1440" "64 2560 1440" "16 1600 1200" "128 1600 1200" "16 5120 2880" "6 4096 4096" "16 3000 2000" "32 3000 2000" "24 1280 720" "128 1280 720" "192 1280 720" \
                "3 7680 4320" "192 1000 1000" "256 640 480" "24 1000 1000" ;;
    sep-rows) printf '%s\n' "4 3840 2160" "8 3840 2160" "16 3840 2160" "32 3840 2160" "16 1920 1080" "32 1920 1080" "64 1920 1080" "128 1920 1080" "16 2560 1440" "64 2560 1440" "64 1280 720" "256 1280 720" ;;
    video)    for S in "1280 720" "1920 1080" "2560 1440" "3840 2160" "1600 1200" "5120 2880" "1000 1000" "3000 2000" "7680 4320" "640 480"; do set -- $S
                for P in 1 2 3 4 6 8 12 16 24 32 48 64 96 128 192 256; do if [ $(( $1 * $2 * P )) -le 1100000000 ] && [ $(( $1 * $2 * P )) -ge 4000000 ]; then echo "$P $1 $2"; fi; done; done ;;
    *)        cat "$SET" ;;
  esac
}
KIND=${WHAT%%:*}; ARG=${WHAT#*:}
{
echo "# sweep_variants.sh set=$SET modes=$MODES what=$WHAT rounds=$ROUNDS map=$MAP"
for M in ${MODES//,/ }; do
  shapes | while read P W H; do
    case "$KIND" in
      variants) timeout 600 python3 tools/ab.py $P $W $M 0 $ARG $ROUNDS $MAP $H ;;
      libs)     tools/ab_libs.sh "$ARG" $P $W $M $MAP 0 $ROUNDS $H ;;
      rows)     for R in ${ARG//,/ }; do timeout 300 python3 tools/ab.py $P $W $M $R 0 $ROUNDS $MAP $H | tail -1 | sed "s/^/mode $M $P x ${W}x$H rows $R: /; s/variant 0: //; s/ssim.*//"; done ;;
      *)        echo "unknown kind $KIND"; exit 2 ;;
    esac
  done
done
} > "$OUT/sweep.txt" 2>&1
tail -5 "$OUT/sweep.txt"
