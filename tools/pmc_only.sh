#!/bin/bash
# the PMC pass of tools/collect_profiles.sh by itself + one short bench line:   bash tools/pmc_only.sh r02
R=${1:-r02}
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/pmc -- python $ROOT/bench.py --layers 8 --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-extras > $ROOT/gpurun_out/pmc.log 2>&1
cd $ROOT
python tools/prof_summary.py --pmc gpurun_out/pmc --out gpurun_out/$R --note "MI355X, round 2 final build, 8-block DeepSeek-V3 Q2_K, MHA path (PMC pass of its own)" > gpurun_out/${R}_pmc_summary.log 2>&1
rm -rf gpurun_out/pmc
python bench.py --steps 32 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | grep '^{' > gpurun_out/${R}_bench_short.json
