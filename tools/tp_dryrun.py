"""SURVEY 8 row f-4 (tensor parallelism), dry run on ONE GPU: what would rank 0 of a W-GPU row-split engine compute per token?

Every GEMV of a DeepSeek-V3 Q2_K token is launched ALONE (include/dsk.h dsk_bench_gemv: rotating weight sets, HBM-resident)
with the row range dsk_tp_rows gives rank 0 at world 1 / 2 / 4 / 8 (replicated GEMVs and the experts split by
OUTPUT rows, attention by heads).  The per-launch times are summed over the launches of a token.  This is an ESTIMATE from
isolated launches, not an engine mode: in the model a launch is ~10-15 % slower than alone (cold descriptors, activations
written by other CUs), and the exchanges (6 all-gathers per MoE block over xGMI) are NOT included - their byte counts are
printed next to the compute so that DESIGN.md can price them.  No scaling curve is measured anywhere in this repository.

    python tools/tp_dryrun.py > profiles/r03_tp_dryrun.json
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))
import dsk  # noqa: E402

Q2K = 3
DIM, H, HD, VD, NOPE, LORA, QLORA, ROPE = 7168, 128, 192, 128, 128, 512, 1536, 64
MI, K, E, HIDDEN, VOCAB = 2048, 8, 256, 18432, 129280
N_DENSE, N_MOE = 3, 58


def rows_of(rows, unit, world):
    f = dsk.lib().dsk_tp_rows
    f.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] * 2
    r0, n = C.c_int(), C.c_int()
    dsk.check(f(rows, unit, world, 0, C.byref(r0), C.byref(n)))
    return n.value


def main():
    ctx = dsk.Ctx(0)
    out = {"what": "rank 0's GEMV share per token at world W, each launch alone (dsk_bench_gemv); exchanges not included", "worlds": {}}
    for W in (1, 2, 4, 8):
        def g(rows, n, nt=1, kind=0, act=2, unit=256):
            r = rows_of(rows, unit, W)
            us, nb = ctx.bench_gemv(Q2K, max(r, 1), n, nt, kind, act, 0, 0, 0, 0, 30)
            return round(us, 2), r
        L = {}
        L["qkv_a (wq_a || wkv_a rows)"] = g(QLORA + 576, DIM, unit=64)
        L["q_b || kv_b rows of H/W heads (fused with attention in the engine: one workgroup per head, ~15.9 us at any W)"] = g(H * (HD + NOPE + VD), QLORA, unit=HD + NOPE + VD)
        L["wo"] = g(DIM, H * VD, act=0)
        L["shared w1/w3"] = g(MI, DIM, kind=1)
        L["experts w1/w3 (8 slots, rows of every selected expert split)"] = g(MI, DIM, nt=8, kind=1, act=0)
        L["experts + shared w2, combine"] = g(DIM, MI, nt=9, kind=3, act=1)
        L["dense w1/w3"] = g(HIDDEN, DIM, kind=1)
        L["dense w2"] = g(DIM, HIDDEN, act=1)
        L["lm_head"] = g(VOCAB, DIM, unit=1)
        router_us = round(ctx.bench_router(E, DIM, 8, 0, 30), 2)  # replicated: 7.3 MB of F32 rows + the gate on every rank
        attn_us = 15.9  # per-head launch: measured in the model (profiles/r03_bench_full.json attn_mha); one workgroup per head at any W
        moe_block = (L["qkv_a (wq_a || wkv_a rows)"][0] + attn_us + L["wo"][0] + router_us + L["shared w1/w3"][0] +
                     L["experts w1/w3 (8 slots, rows of every selected expert split)"][0] + L["experts + shared w2, combine"][0])
        dense_block = L["qkv_a (wq_a || wkv_a rows)"][0] + attn_us + L["wo"][0] + L["dense w1/w3"][0] + L["dense w2"][0]
        token_us = N_MOE * moe_block + N_DENSE * dense_block + L["lm_head"][0]
        # exchanges of the design, bytes each rank RECEIVES per all-gather (f32 unless noted)
        gathers = {"q_a || kv_a (2112 f32)": 2112 * 4, "attention output as Q8_K (16384 codes + sums + scales)": 16384 + 16384 // 16 * 2 + 16384 // 256 * 4,
                   "x after wo (7168 f32)": DIM * 4, "hidden vectors as Q8_K (9 x 2048)": 9 * (2048 + 2048 // 16 * 2 + 2048 // 256 * 4),
                   "x after the FFN (7168 f32)": DIM * 4}
        out["worlds"][str(W)] = dict(launch_us={k: v[0] for k, v in L.items()}, rows_rank0={k: v[1] for k, v in L.items()},
                                     router_replicated_us=router_us, attention_us_assumed=attn_us,
                                     moe_block_us=round(moe_block, 2), dense_block_us=round(dense_block, 2), token_compute_ms=round(token_us / 1e3, 4),
                                     gathers_per_moe_block=len(gathers), gather_bytes=gathers,
                                     gather_bytes_per_token=int(sum(gathers.values()) * (N_MOE + N_DENSE) * (W - 1) / max(W, 1)))
    w1 = out["worlds"]["1"]["token_compute_ms"]
    for W, d in out["worlds"].items():
        d["compute_speedup_vs_world1"] = round(w1 / d["token_compute_ms"], 3)
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
