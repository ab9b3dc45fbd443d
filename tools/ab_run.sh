#!/bin/bash
# tools/ab_run.sh NAME...: a short bench (and, with AB_TL=1, the timeline of the routed-expert launch; with AB_TEST=1, the
# fused-launch tests) for each A/B build of tools/ab_build.sh
cd /root/repo
for v in "$@"; do
  export DSK_LIB=/root/repo/deepseek.cpp_amd/_ab/libdsk_$v.so
  echo "=== $v" > gpurun_out/ab_$v.log
  if [ -n "$AB_TL" ]; then timeout 120 python tools/moe_timeline.py 2>&1 | grep -v amdgpu.ids | head -9 >> gpurun_out/ab_$v.log; fi
  timeout 200 python bench.py --steps 32 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab_$v.log
  if [ -n "$AB_TEST" ]; then timeout 600 python -m pytest tests/test_fused_moe_gpu.py -x -q 2>&1 | tail -3 >> gpurun_out/ab_$v.log; fi
done
