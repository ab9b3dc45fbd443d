#!/bin/bash
# Where do the SIMDs spend their time INSIDE the model?  SQ counters (rocprofv3 --pmc, two passes of 8 counters each, counters
# only: no other trace domain) over an eager 8-block DeepSeek-V3 Q2_K decode, summarised per kernel of the token.
#   bash tools/pmc_sq.sh r04  ->  gpurun_out/r04_pmc_sq.txt
# Reading: SQ_WAVE_CYCLES = wave-resident time in quad-cycles summed over waves; SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.0 means one
# quad-cycle (4 clocks) per wave VALU instruction; with 4 waves per SIMD (16-wave workgroups, one per CU) a SIMD's VALU-busy share
# is 4 x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES.  SQ_WAIT_ANY = parked on s_waitcnt / barrier; SQ_WAIT_INST_ANY = waiting to issue.
R=${1:-r04}
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
ARGS="--layers 8 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extras ${PMC_BENCH_ARGS}"
cd /tmp
timeout 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace -d $OUT/p1 -- python $ROOT/bench.py $ARGS < /dev/null > $OUT/p1.log 2>&1
timeout 180 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace -d $OUT/p2 -- python $ROOT/bench.py $ARGS < /dev/null > $OUT/p2.log 2>&1
cd $ROOT
python3 - "$OUT" > gpurun_out/${R}_pmc_sq.txt <<'PY'
import collections, glob, re, sqlite3, sys
OUT = sys.argv[1]
print("# SQ counters per launch (mean over the launches of an eager 8-block decode), rocprofv3 --pmc, MI355X; tools/pmc_sq.sh")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for db in sorted(glob.glob(OUT + "/p*/**/*.db", recursive=True)):
    con = sqlite3.connect(db)
    try:
        rows = con.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    except Exception as e:  # schema differences between rocprofv3 builds
        print("# could not read", db, e); continue
    for k, c, v in rows:
        name = re.sub(r"\(.*", "", k)
        name = re.sub(r"^void ", "", name)
        agg[name][c].append(v)
for name in sorted(agg, key=lambda n: -sum(agg[n].get("SQ_WAVE_CYCLES", [0]))):
    c = {k: sum(v) / len(v) for k, v in agg[name].items()}
    n = len(next(iter(agg[name].values())))
    if c.get("SQ_WAVE_CYCLES", 0) < 1e4: continue
    print(f"== {name}   ({n} launches)")
    for k in sorted(c): print(f"   {k:28s} {c[k]:14.0f}")
    if "SQ_WAVE_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
        wc = c["SQ_WAVE_CYCLES"]
        print(f"   quad-cycles per VALU instruction              {c['SQ_ACTIVE_INST_VALU'] / max(1.0, c['SQ_INSTS_VALU']):.3f}")
        print(f"   VALU-active share of wave time                {c['SQ_ACTIVE_INST_VALU'] / wc:.3f}   (x waves per SIMD = the SIMD's VALU-busy share; 4 for a 16-wave workgroup per CU)")
        print(f"   wave time parked (s_waitcnt / barrier)        {c['SQ_WAIT_ANY'] / wc:.3f}")
        print(f"   wave time waiting to issue                    {c['SQ_WAIT_INST_ANY'] / wc:.3f}")
        if "SQ_INSTS_VMEM_RD" in c: print(f"   VALU instructions per VMEM read instruction   {c['SQ_INSTS_VALU'] / max(1.0, c['SQ_INSTS_VMEM_RD']):.1f}")
PY
rm -rf $OUT/p1 $OUT/p2
head -150 gpurun_out/${R}_pmc_sq.txt
