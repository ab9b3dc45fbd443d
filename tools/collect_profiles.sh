#!/bin/bash
# Round evidence, one gpurun call on one box (profiles/README.md): bench line, rocprofv3 kernel traces (MHA, MLA, kv_len 4096),
# a PMC pass of its own for HBM traffic, SQ counters in situ, the op-level GEMV table, the in-kernel timelines, the probes.
#   bash tools/collect_profiles.sh r05        (every step bounded by `timeout`, nothing reads stdin)
R=${1:-r06}
N="round ${R#r0} final build"
mkdir -p gpurun_out profiles
export TMPDIR=/tmp
ROOT=$PWD
T="timeout 300"
$T python bench.py --steps 64 --warmup 8 < /dev/null > gpurun_out/${R}_bench_full.log 2>&1
grep '^{' gpurun_out/${R}_bench_full.log | tail -1 > gpurun_out/${R}_bench_full.json
$T python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras < /dev/null 2>&1 | grep '^{' | tail -1 > gpurun_out/${R}_bench_short.json
$T python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --opt q2k_tiles=0 < /dev/null 2>&1 | grep '^{' | tail -1 > gpurun_out/${R}_bench_short_planes.json
$T python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --opt q2k_tiles=2 < /dev/null 2>&1 | grep '^{' | tail -1 > gpurun_out/${R}_bench_short_alltiles.json
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/trace_mha -- python $ROOT/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extras < /dev/null > $ROOT/gpurun_out/trace_mha.log 2>&1
timeout 240 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/trace_mla -- python $ROOT/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extras --attn mla < /dev/null > $ROOT/gpurun_out/trace_mla.log 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/pmc -- python $ROOT/bench.py --layers 8 --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-extras < /dev/null > $ROOT/gpurun_out/pmc.log 2>&1
cd $ROOT
$T python tools/prof_summary.py --trace gpurun_out/trace_mha --pmc gpurun_out/pmc --out gpurun_out/$R --note "MI355X, $N, full 61-block DeepSeek-V3 Q2_K, MHA path" < /dev/null > gpurun_out/${R}_summary.log 2>&1
$T python tools/prof_summary.py --trace gpurun_out/trace_mla --out gpurun_out/${R}_mla --note "MI355X, $N, full 61-block DeepSeek-V3 Q2_K, MLA path" < /dev/null >> gpurun_out/${R}_summary.log 2>&1
$T python tools/kbench.py < /dev/null 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_kbench.txt
$T python tools/timeline.py < /dev/null 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_timeline_mha.txt
$T python tools/moe_timeline.py < /dev/null 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_timeline_moe.txt
# round 5: the floor of a five-launch block on this box; the prompt phase (dsk_hydrate) on the full model and its kernel trace
timeout 120 tools/_build/block_floor 58 < /dev/null > gpurun_out/${R}_block_floor.txt 2>&1
# (round 6: the engine's DEFAULT options - the model bench.py times for decode; the batched path on tile copies of the plane matrices;
# _alltiles: q2k_tiles = 2, the layout whose batched path is bit-identical to the per-token loop)
$T python tools/hydrate_bench.py --P 16,64,128,256,512 < /dev/null 2>&1 | grep '^{' | tail -1 > gpurun_out/${R}_hydrate.json
$T python tools/hydrate_bench.py --P 64,512 --opt q2k_tiles=2 < /dev/null 2>&1 | grep '^{' | tail -1 > gpurun_out/${R}_hydrate_alltiles.json
# uniform routing instead of the synthetic model's skewed one: a measurement knob of -DDSK_AB builds only (tools/ab_build.sh ab "")
AB=deepseek.cpp_amd/_ab/libdsk_ab.so
[ -f $AB ] && DSK_LIB=$AB DSK_HYD_ROUTE_SEED=7 $T python tools/hydrate_bench.py --P 16,64,128,256,512 < /dev/null 2>&1 | grep '^{' | tail -1 > gpurun_out/${R}_hydrate_uniform.json
# (MLA through the matrix-core regime, round 6: prompts past position 319 batch; 256 tokens at positions 512-767)
$T python tools/hydrate_bench.py --attn mla --P 16,64,128,256,512,1024 < /dev/null 2>&1 | grep '^{' | tail -1 > gpurun_out/${R}_hydrate_mla.json
$T python tools/hydrate_bench.py --attn mla --P 256 --pos0 512 --no-loop < /dev/null 2>&1 | grep '^{' | tail -1 > gpurun_out/${R}_hydrate_mla_pos512.json
[ -f $AB ] && DSK_LIB=$AB DSK_HYD_ROUTE_SEED=7 $T python tools/hydrate_bench.py --attn mla --P 64,128 < /dev/null 2>&1 | grep '^{' | tail -1 > gpurun_out/${R}_hydrate_mla_uniform.json
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/trace_hyd -- python $ROOT/tools/hydrate_bench.py --P 64 --layers 8 --reps 1 --no-loop < /dev/null > $ROOT/gpurun_out/trace_hyd.log 2>&1
cd $ROOT
$T python tools/prof_summary.py --trace gpurun_out/trace_hyd --out gpurun_out/${R}_hydrate --note "MI355X, $N, dsk_hydrate of 64 tokens (warm-up + one timed call) on 8 blocks (3 dense + 5 MoE) of DeepSeek-V3 Q2_K, tile records everywhere" < /dev/null >> gpurun_out/${R}_summary.log 2>&1
rm -rf gpurun_out/trace_hyd
# DeepSeek-V2-Lite (BASELINE configs C3) in both Q2_K layouts; the Q8_K hand-over of the fused expert launch against f32 hidden vectors
$T python bench.py --model v2lite --steps 32 --warmup 4 --no-cpu-baseline --no-extras < /dev/null 2>&1 | grep '^{' | tail -1 > gpurun_out/${R}_bench_v2lite.json
$T python bench.py --model v2lite --steps 32 --warmup 4 --no-cpu-baseline --no-extras --opt q2k_tiles=0 < /dev/null 2>&1 | grep '^{' | tail -1 > gpurun_out/${R}_bench_v2lite_planes.json
rm -rf gpurun_out/trace_mha gpurun_out/trace_mla gpurun_out/pmc gpurun_out/pmc_sq
# round 6: the scalar-path publish probe; the experimental schedules of the fused expert launch next to the shipped one (A/B lines + stamps)
[ -x tools/_build/scalar_store_probe ] && timeout 120 tools/_build/scalar_store_probe < /dev/null 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_scalar_store_probe.txt
( $T python tools/moe_ab.py; $T python tools/moe_ab.py --opt moe_pipe=1; $T python tools/moe_ab.py --opt moe_pipe=2; $T python tools/moe_ab.py --attn mla; $T python tools/moe_ab.py --attn mla --opt ride_kvwrite=0 ) < /dev/null 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_ab_moe_pipe.txt
# round 6, late: the launches that request their first weights ahead of the staging (option gemv_ahead: bits 1 first-stage / MLA second stage, 2 wo)
( for v in 0 1 3; do $T python tools/moe_ab.py --opt gemv_ahead=$v; done; $T python tools/moe_ab.py --attn mla --opt gemv_ahead=0; $T python tools/moe_ab.py --attn mla --opt gemv_ahead=2; $T python tools/moe_ab.py --attn mla --opt gemv_ahead=3 ) < /dev/null 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_ab_gemv_ahead.txt
$T python tools/timeline.py --opt gemv_ahead=0 < /dev/null 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_timeline_mha_noahead.txt
$T python tools/moe_timeline.py --layers 61 --opt moe_pipe=1 < /dev/null 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_timeline_moe_pipe1.txt
$T python tools/moe_timeline.py --layers 61 --opt moe_pipe=2 < /dev/null 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_timeline_moe_pipe2.txt
# the whole GPU test-suite and the smoke entry point on the same sources, same box
( timeout 900 python -m pytest tests -q -m gpu < /dev/null; timeout 300 python __graft_entry__.py smoke < /dev/null ) > gpurun_out/${R}_gputests.log 2>&1
tail -3 gpurun_out/${R}_gputests.log
ls gpurun_out | grep "^${R}_" | head -40
