"""ms per decode step of the full DeepSeek-V3 Q2_K model at given context lengths (6 timed steps after 2 of warm-up each).
    python tools/kv_time.py mla 128 256 512 768 1024 2048 4096      (MLA_FLASH_MIN=... in the environment of THIS script moves the MLA regime switch: option "mla_flash_min")"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))
import dsk
from tools import synth

mla = len(sys.argv) > 1 and sys.argv[1] == "mla"
kvs = [int(a) for a in sys.argv[2:]] or [128, 1024, 4096]
c = synth.preset("v3", "q2_k", mla, max_seq_len=4200)
ctx = dsk.Ctx(0)
fm = os.environ.get("MLA_FLASH_MIN")
M = dsk.Model(ctx, c, None, synth_seed=0, options={"mla_flash_min": int(fm)} if fm else None)
out = {}
for kv in kvs:
    for i in range(2):
        M.forward_nocopy(100 + i, kv - 1 + i)
    ctx.sync() if hasattr(ctx, "sync") else None
    t0 = time.perf_counter()
    for i in range(6):
        M.forward(100 + i, kv + 1 + i)
    out[kv] = round((time.perf_counter() - t0) / 6 * 1e3, 4)
print({"attn": "mla" if mla else "mha", "flash_min": fm or "default", "ms_per_step": out})
M.close(); ctx.close()
