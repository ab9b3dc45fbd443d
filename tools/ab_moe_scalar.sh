#!/bin/bash
# A/B of the fused expert launch with the chain first, a first look, scalar polls and the copies in front of the late W2 requests
# (kernels_moe_tile.hip), against the previous commit's kernel.  Builds, before the call:
#   for e in 3 4 5; do bash tools/ab_build.sh S1E$e "-DMOE_T_EARLY=$e -DMOE_TL7=1"; done;  libdsk_base.so = the previous commit's sources
O=gpurun_out/r04_ab_moe_scalar.txt
mkdir -p gpurun_out; : > $O
for v in base S1E4 S1E3 S1E5 base S1E4; do
  DSK_LIB=deepseek.cpp_amd/_ab/libdsk_$v.so timeout 120 python tools/moe_ab.py --steps 32 < /dev/null 2>&1 | grep -v amdgpu.ids >> $O
done
for v in S1E4 S1E3; do
  echo "== timeline $v" >> $O
  DSK_LIB=deepseek.cpp_amd/_ab/libdsk_$v.so timeout 120 python tools/moe_timeline.py < /dev/null 2>&1 | grep -v amdgpu.ids | head -13 >> $O
done
DSK_LIB=deepseek.cpp_amd/_ab/libdsk_S1E4.so timeout 200 python -m pytest tests/test_tiles_gpu.py tests/test_fused_moe_gpu.py -x -q < /dev/null 2>&1 | tail -3 >> $O
cat $O
