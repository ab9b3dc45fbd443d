cd /tmp; export TMPDIR=/tmp
for kv in 1024 4096; do
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/trace_mla_kv$kv -- python /root/repo/tools/kv_trace.py $kv mla > /root/repo/gpurun_out/trace_mla_kv$kv.log 2>&1
cd /root/repo; python tools/prof_summary.py --trace gpurun_out/trace_mla_kv$kv --out gpurun_out/r02_mla_kv$kv --note "MI355X, DeepSeek-V3 Q2_K MLA, 8 decode steps at kv_len $kv (tools/kv_trace.py $kv mla)" > /dev/null 2>&1; rm -rf gpurun_out/trace_mla_kv$kv; cd /tmp
done
