cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in 0 1 3; do
timeout 300 python tools/moe_ab.py --opt gemv_ahead=$v 2>&1 | grep -v amdgpu.ids
done
done
timeout 300 python tools/moe_ab.py --attn mla --opt gemv_ahead=0 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/moe_ab.py --attn mla --opt gemv_ahead=3 2>&1 | grep -v amdgpu.ids
timeout 200 python tools/timeline.py --opt gemv_ahead=3 2>&1 | grep -A8 "wo (gemv)"
