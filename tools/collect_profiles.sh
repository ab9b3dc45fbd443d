#!/bin/bash
# Round evidence, one gpurun call on one box (profiles/README.md): bench line, rocprofv3 kernel traces (MHA, MLA, kv_len 4096),
# a PMC pass of its own for HBM traffic, the op-level GEMV table, the in-kernel timelines, the reduced-depth C1 CPU line.   bash tools/collect_profiles.sh r03
R=${1:-r03}
N="round ${R#r0} final build"
mkdir -p gpurun_out profiles
export TMPDIR=/tmp
ROOT=$PWD
python bench.py --steps 64 --warmup 8 > gpurun_out/${R}_bench_full.log 2>&1
grep '^{' gpurun_out/${R}_bench_full.log | tail -1 > gpurun_out/${R}_bench_full.json
for cfg in "mla --attn mla" "v2lite --model v2lite" "v2lite_f8 --model v2lite --quant f8e5m2"; do
  set -- $cfg; name=$1; shift
  python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-extras "$@" > gpurun_out/${R}_bench_$name.log 2>&1
  grep '^{' gpurun_out/${R}_bench_$name.log | tail -1 > gpurun_out/${R}_bench_$name.json
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/trace_mha -- python $ROOT/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extras > $ROOT/gpurun_out/trace_mha.log 2>&1
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/trace_mla -- python $ROOT/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extras --attn mla > $ROOT/gpurun_out/trace_mla.log 2>&1
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/trace_kv4096 -- python $ROOT/tools/kv_trace.py 4096 > $ROOT/gpurun_out/trace_kv4096.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/pmc -- python $ROOT/bench.py --layers 8 --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-extras > $ROOT/gpurun_out/pmc.log 2>&1
cd $ROOT
python tools/prof_summary.py --trace gpurun_out/trace_mha --pmc gpurun_out/pmc --out gpurun_out/$R --note "MI355X, $N, full 61-block DeepSeek-V3 Q2_K, MHA path" > gpurun_out/${R}_summary.log 2>&1
python tools/prof_summary.py --trace gpurun_out/trace_mla --out gpurun_out/${R}_mla --note "MI355X, $N, full 61-block DeepSeek-V3 Q2_K, MLA path" >> gpurun_out/${R}_summary.log 2>&1
python tools/prof_summary.py --trace gpurun_out/trace_kv4096 --out gpurun_out/${R}_kv4096 --note "MI355X, $N, DeepSeek-V3 Q2_K MHA, 6 decode steps at kv_len 4096 (tools/kv_trace.py)" >> gpurun_out/${R}_summary.log 2>&1
python tools/kbench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_kbench.txt
for w in 2 4 8; do python bench.py --dry-shard 0/$w --steps 20 --warmup 4 --no-cpu-baseline --no-extras 2>&1 | grep '^{' > gpurun_out/${R}_dry_shard_0of$w.json; done
python tools/timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_timeline_mha.txt
python tools/timeline.py --attn mla 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_timeline_mla.txt
python tools/timeline.py --attn mla --kv 4096 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_timeline_mla_kv4096.txt
python tools/moe_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_timeline_moe.txt
python tools/tp_dryrun.py 2> gpurun_out/${R}_tp_dryrun.log | grep -v amdgpu.ids > gpurun_out/${R}_tp_dryrun.json
bash tools/pmc_gemv.sh $R > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ll_probe tools/ll_probe.hip > /dev/null 2>&1 && timeout 60 /tmp/ll_probe > gpurun_out/${R}_ll_probe.txt 2>&1
if [ -z "$SKIP_C1" ]; then python tools/cpu_c1.py > gpurun_out/${R}_cpu_c1.json 2> gpurun_out/${R}_cpu_c1.log; fi
rm -rf gpurun_out/trace_mha gpurun_out/trace_mla gpurun_out/trace_kv4096 gpurun_out/pmc
ls -la gpurun_out | tail -30
