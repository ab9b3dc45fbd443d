#!/bin/bash
# A/B builds of the engine library: tools/ab_build.sh NAME "-DKNOB=1 ..."  ->  deepseek.cpp_amd/_ab/libdsk_NAME.so
# (run the variant with DSK_LIB=deepseek.cpp_amd/_ab/libdsk_NAME.so; _ab/ is git-ignored and travels with gpurun)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/deepseek.cpp_amd/_ab
make -s -j8 -C $ROOT/deepseek.cpp_amd/csrc BUILD=$ROOT/deepseek.cpp_amd/_ab/build_$1 OUT=$ROOT/deepseek.cpp_amd/_ab/libdsk_$1.so EXTRA="-DDSK_AB $2" 2>&1 | grep -v "warning\|amdgpu.ids\|dot6-insts\|\^\|^ *[0-9]* |" || true
ls -la $ROOT/deepseek.cpp_amd/_ab/libdsk_$1.so
