O=gpurun_out/r04_ab_moe_stagewave.txt
mkdir -p gpurun_out; : > $O
for v in base CF0SW0 CF1SW0 CF0SW1 CF1SW1 base CF1SW1 CF1SW0; do
  f=deepseek.cpp_amd/_ab/libdsk_$v.so
  DSK_LIB=$f timeout 120 python tools/moe_ab.py --steps 32 < /dev/null 2>&1 | grep -v amdgpu.ids >> $O
done
for v in CF0SW0 CF1SW0 CF0SW1 CF1SW1; do
  echo "== timeline $v" >> $O
  DSK_LIB=deepseek.cpp_amd/_ab/libdsk_$v.so timeout 120 python tools/moe_timeline.py < /dev/null 2>&1 | grep -v amdgpu.ids | head -13 >> $O
done
DSK_LIB=deepseek.cpp_amd/_ab/libdsk_CF1SW1.so timeout 200 python -m pytest tests/test_tiles_gpu.py tests/test_fused_moe_gpu.py -x -q < /dev/null 2>&1 | tail -3 >> $O
cat $O
