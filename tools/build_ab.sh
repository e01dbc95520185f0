#!/bin/bash
# Builds a variant of the product library for tools/ab_libs.sh: build/ab/lib<NAME>.so = the kernels compiled with extra
# flags + the current C ABI / drop-in objects (make lib first).   usage: tools/build_ab.sh NAME [extra hipcc flags...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Issim_amd/csrc -Wall -Wno-unused-function "$@" -c ssim_amd/csrc/ssim_kernels.hip -o build/ab/k_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=ssim_amd/csrc/exports.map -o build/ab/lib$NAME.so build/ab/k_$NAME.o build/obj/ssim_hip_abi.o build/obj/ssim_dropin.o
echo "build/ab/lib$NAME.so"
