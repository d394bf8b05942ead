#!/bin/bash
# Build a variant of liblstc_hip.so + gemm_check into build/<name>/ with extra -D flags on ONE source file:
#   tools/build_variant.sh nt gemm_bf16p '-DP1_STORE_MOD=" nt"'
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p build/$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off "$@" -c lstc_vad_amd/csrc/$src.hip -o build/$name/$src.o 2>&1 | grep -i "error" || true
objs=$(ls lstc_vad_amd/csrc/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/$name/liblstc_hip.so $objs build/$name/$src.o
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude tools/gemm_check.cpp -o build/$name/gemm_check -Lbuild/$name -llstc_hip -Wl,-rpath,'$ORIGIN' 2>&1 | grep -i "error" || true
ls -la build/$name/
