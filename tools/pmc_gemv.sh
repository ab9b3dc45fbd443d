#!/bin/bash
# Where do the SIMDs of a Q2_K GEMV launch spend their time?  SQ counters (rocprofv3 --pmc, two passes of 8 counters, counters
# only: no other trace domain) for the classifier (129280 x 7168, 16 lanes per row) and wo (7168 x 16384, 64 lanes per row),
# each launched alone 20 times on rotating weight sets.   bash tools/pmc_gemv.sh r03  ->  gpurun_out/r03_pmc_gemv.txt
# Reading: SQ_WAVE_CYCLES = wave-resident time in quad-cycles summed over waves, SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.0 means
# one quad-cycle (4 clocks) per wave VALU instruction; a SIMD holds 4 waves here (16 waves per CU), so the SIMDs' VALU-busy
# share is 4 x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES.
R=${1:-r03}
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_gemv
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for shape in lm_head wo; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace -d $OUT/p1_$shape -- python $ROOT/tools/pmc_gemv.py $shape > $OUT/p1_$shape.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace -d $OUT/p2_$shape -- python $ROOT/tools/pmc_gemv.py $shape > $OUT/p2_$shape.log 2>&1
done
cd $ROOT
python3 - "$OUT" > gpurun_out/${R}_pmc_gemv.txt <<'PY'
import collections, glob, sqlite3, subprocess, sys
OUT = sys.argv[1]
print("# SQ counters per launch (mean over the launches seen), rocprofv3 --pmc, MI355X; tools/pmc_gemv.sh")
rows = {}
for db in sorted(glob.glob(OUT + "/p*_*/**/*.db", recursive=True)):
    shape = db[len(OUT) + 1:].split("/")[0][3:]
    con = sqlite3.connect(db)
    agg = collections.defaultdict(list)
    for k, c, v in con.execute("select kernel_name, counter_name, value from counters_collection"):
        if "gemv_kernel" in k: agg[c].append(v)
    rows.setdefault(shape, {}).update({c: sum(v) / len(v) for c, v in agg.items()})
for shape, c in rows.items():
    print(f"== {shape}")
    for k in sorted(c): print(f"   {k:24s} {c[k]:14.0f}")
    if "SQ_WAVE_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
        print(f"   quad-cycles per VALU instruction          {c['SQ_ACTIVE_INST_VALU'] / max(1.0, c['SQ_INSTS_VALU']):.3f}")
        print(f"   SIMD VALU-busy share (4 waves per SIMD)   {4 * c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.2f}")
        print(f"   wave time waiting for anything            {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f}")
        print(f"   wave time waiting to issue                {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f}")
PY
for shape in lm_head wo; do grep -h "us" $OUT/p1_$shape.log | tail -1 >> gpurun_out/${R}_pmc_gemv.txt; done
rm -rf $OUT
cat gpurun_out/${R}_pmc_gemv.txt
