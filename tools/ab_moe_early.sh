#!/bin/bash
# A/B of MOE_T_EARLY (kernels_moe_tile.hip): how many of a wave's register steps of W2 are requested BEFORE the hand-off wait.
# Builds (here, before the call):  for e in 0 1 2 3 4 5 8; do bash tools/ab_build.sh E$e "-DMOE_T_EARLY=$e"; done
# and libdsk_base.so = the previous commit's sources.   bash tools/ab_moe_early.sh  ->  gpurun_out/r04_ab_moe_early.txt
O=gpurun_out/r04_ab_moe_early.txt
mkdir -p gpurun_out
: > $O
for v in base E8 E0 E1 E2 E3 E4 E5 base E2 E3; do
  f=deepseek.cpp_amd/_ab/libdsk_$v.so
  [ -f $f ] || continue
  DSK_LIB=$f timeout 120 python tools/moe_ab.py --steps 32 < /dev/null 2>&1 | grep -v amdgpu.ids >> $O
done
for v in base E2 E3 E4; do
  f=deepseek.cpp_amd/_ab/libdsk_$v.so
  [ -f $f ] || continue
  echo "== timeline $v" >> $O
  DSK_LIB=$f timeout 120 python tools/moe_timeline.py < /dev/null 2>&1 | grep -v amdgpu.ids | head -9 >> $O
done
cat $O
