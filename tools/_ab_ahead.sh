cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 300 python tools/moe_ab.py --attn mla --opt gemv_ahead=0 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/moe_ab.py --attn mla --opt gemv_ahead=2 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/moe_ab.py --attn mla --opt gemv_ahead=3 2>&1 | grep -v amdgpu.ids
done
timeout 300 python tools/moe_ab.py --opt gemv_ahead=3 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_teacher_forced_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3
