#!/bin/bash
# the shipped plan() (libN1024: cap 1024, balanced rule against the 512-row strips) against round 5's earlier plan() (libC512) on the shapes where the two interact
cd "$(dirname "$0")/.."
for P in 16 24 48 64 96 128 256; do tools/ab_libs.sh "C512 N1024" $P 1920 0 0 0 2 1080; done
for P in 64 128; do tools/ab_libs.sh "C512 N1024" $P 1920 1 0 0 2 1080; done
for P in 3 32; do tools/ab_libs.sh "C512 N1024" $P 4096 0 0 0 2; done
tools/ab_libs.sh "C512 N1024" 32 4096 1 0 0 2
tools/ab_libs.sh "C512 N1024" 8 8192 0 1 0 2
