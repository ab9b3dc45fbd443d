#!/bin/bash
# A/B of the fused expert launch's phase-B sequencing (kernels_moe_tile.hip): MOE_CHAIN_FIRST (the hand-over chain completes before any
# W2 request) x MOE_T_EARLY (register steps of W2 requested before the hand-off wait).  Builds, before the call:
#   for e in 2 3 4 5 8; do bash tools/ab_build.sh C1E$e "-DMOE_T_EARLY=$e -DMOE_CHAIN_FIRST=1 -DMOE_TL7=1"; done
#   bash tools/ab_build.sh C0E8 "-DMOE_T_EARLY=8 -DMOE_CHAIN_FIRST=0 -DMOE_TL7=1";  libdsk_base.so = the previous commit's sources
# bash tools/ab_moe_early.sh [OUT]  ->  gpurun_out/r04_ab_moe_chain.txt
O=${1:-gpurun_out/r04_ab_moe_chain.txt}
mkdir -p gpurun_out
: > $O
for v in base C0E8 C1E8 C1E5 C1E4 C1E3 C1E2 base C1E4 C1E3; do
  f=deepseek.cpp_amd/_ab/libdsk_$v.so
  [ -f $f ] || continue
  DSK_LIB=$f timeout 120 python tools/moe_ab.py --steps 32 < /dev/null 2>&1 | grep -v amdgpu.ids >> $O
done
for v in C1E8 C1E5 C1E4 C1E3 C1E2; do
  f=deepseek.cpp_amd/_ab/libdsk_$v.so
  [ -f $f ] || continue
  echo "== timeline $v" >> $O
  DSK_LIB=$f timeout 120 python tools/moe_timeline.py < /dev/null 2>&1 | grep -v amdgpu.ids | head -13 >> $O
done
cat $O
