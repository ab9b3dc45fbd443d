"""Generate tests/golden/*.npz from the UNMODIFIED reference (oracle/_ref/libdskref.so).

Run in the build container (needs /root/reference to have been compiled by oracle/Makefile):
    python tools/make_golden.py
The fixtures pin the oracle (tests/test_oracle_pin.py, CPU) and the HIP path (tests/*_gpu.py) to
outputs of the reference itself; /root/reference is not needed to *use* them.
"""
from __future__ import annotations

import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from tools import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

MODEL_CASES = [  # (preset, quant, use_mla, seed)
    ("tiny_v3", "q2_k", False, 7), ("tiny_v3", "q2_k", True, 7), ("tiny_v3", "q3_k", False, 8),
    ("tiny_v3", "q3_k", True, 8), ("tiny_v3", "f8e5m2", False, 9), ("tiny_v3", "f8e5m2", True, 9),
    ("tiny_v2lite", "q2_k", False, 10), ("tiny_v2lite", "f8e5m2", False, 11), ("tiny_v2lite", "fp16", False, 12),
    ("tiny_v2lite", "fp32", False, 13),
]
TOKENS = [5, 17, 300, 44, 9, 1000, 3, 77]
# independent single-token trials at pos 0 (kv_len = 1: no dependence on earlier state)
TOKENS0 = [1, 2, 8, 21, 34, 55, 89, 144, 233, 377, 610, 987, 64, 128, 256, 511]


def model_sha(T) -> str:
    h = hashlib.sha256()
    for name in sorted(T):
        h.update(name.encode())
        h.update(np.ascontiguousarray(T[name].data).tobytes())
        if T[name].scale is not None:
            h.update(np.ascontiguousarray(T[name].scale).tobytes())
    return h.hexdigest()


def case_name(preset, quant, mla, seed):
    return f"model_{preset}_{quant}_{'mla' if mla else 'mha'}_s{seed}"


def gen_models(R):
    for preset, quant, mla, seed in MODEL_CASES:
        c = synth.preset(preset, quant, mla)
        T = synth.synth_model(c, seed=seed)
        d = tempfile.mkdtemp()
        synth.write_dseek(d, c, T)
        S = R.session(d, c)
        logits, routes_e, routes_w, xs = [], [], [], []
        for pos, t in enumerate(TOKENS):
            t = t % c.vocab_size
            logits.append(S.forward(t, pos, traced=True))
            e, w = S.routing()
            routes_e.append(e)
            routes_w.append(w)
            xs.append(np.stack([S.trace_x(l) for l in range(c.n_layers)]))
        S0 = R.session(d, c)
        logits0, route0 = [], []
        for t in TOKENS0:
            logits0.append(S0.forward(t % c.vocab_size, 0, traced=True))
            route0.append(S0.routing()[0])
        # a fresh session through the unmodified Model::forward must agree bit-for-bit with the traced replay
        S2 = R.session(d, c)
        for pos, t in enumerate(TOKENS[:3]):
            assert np.array_equal(S2.forward(t % c.vocab_size, pos, traced=False), logits[pos])
        np.savez_compressed(os.path.join(GOLD, case_name(preset, quant, mla, seed) + ".npz"),
                            sha=np.frombuffer(bytes.fromhex(model_sha(T)), np.uint8), tokens=np.array(TOKENS),
                            logits=np.stack(logits), route_e=np.stack(routes_e), route_w=np.stack(routes_w),
                            trace_x=np.stack(xs), tokens0=np.array(TOKENS0), logits0=np.stack(logits0),
                            route0_e=np.stack(route0))
        print("model", preset, quant, mla, "ok")


def gen_ops(R):
    rng = np.random.default_rng(20250925)
    out = {}
    # Q8_K activation quantisation incl. edge cases: zero block, tie on |max| with opposite signs, clamp at 127
    x = rng.standard_normal(2048).astype(np.float32) * 3
    x[256:512] = 0.0
    x[512] = 2.5
    x[513] = -2.5
    x[514:768] *= 0.1
    x[768:1024] = np.float32(1e-30)
    out["q8_x"] = x
    out["q8_qs"], out["q8_d"], out["q8_bsums"] = R.q8k_quantize(x)
    # GEMVs on all five quants
    n, d = 1024, 48
    w = (rng.standard_normal((d, n)) / 32).astype(np.float32)
    xv = rng.standard_normal(n).astype(np.float32)
    out["gemv_x"] = xv
    out["w_q2k"] = synth.encode_q2k(w)
    out["w_q3k"] = synth.encode_q3k(w)
    out["y_q2k"] = R.gemv(3, out["w_q2k"], d, n, xv)
    out["y_q3k"] = R.gemv(4, out["w_q3k"], d, n, xv)
    # the reference's own offline quantizer output must decode identically through both paths
    out["w_q2k_ref"] = R.quantize_rows(3, w)
    out["w_q3k_ref"] = R.quantize_rows(4, w)
    out["y_q2k_ref"] = R.gemv(3, out["w_q2k_ref"], d, n, xv)
    out["y_q3k_ref"] = R.gemv(4, out["w_q3k_ref"], d, n, xv)
    out["deq_q2k"] = R.dequant_row(3, out["w_q2k_ref"][5], n)
    out["deq_q3k"] = R.dequant_row(4, out["w_q3k_ref"][5], n)
    w16 = w.astype(np.float16)
    out["w_f16"] = w16.view(np.uint16)
    out["y_f16"] = R.gemv(1, w16, d, n, xv)
    out["w_f32"] = w
    out["y_f32"] = R.gemv(0, w, d, n, xv)
    wbig = (rng.standard_normal((256, n)) / 32).astype(np.float32)
    w8, s8 = synth.encode_f8_blocks(wbig, (128, 128))
    out["w_f8"], out["s_f8"] = w8, s8
    out["y_f8"] = R.gemv(2, w8, 256, n, xv, s8, (128, 128))
    # stacked experts (matmul_expert)
    we = (rng.standard_normal((4, 32, 512)) / 22).astype(np.float32)
    xe = rng.standard_normal(512).astype(np.float32)
    weq = synth.encode_q2k(we.reshape(-1, 512)).reshape(4, 32, -1)
    out["we_q2k"], out["xe"] = weq, xe
    out["ye_q2k"] = np.stack([R.gemv_expert(3, weq, 4, e, 32, 512, xe) for e in range(4)])
    # rmsnorm
    xn = (rng.standard_normal(768) * 2).astype(np.float32)
    wn = (1 + 0.1 * rng.standard_normal(768)).astype(np.float32)
    out["rms_x"], out["rms_w"], out["rms_y"] = xn, wn, R.rmsnorm(xn, wn, 1e-6)
    # moe_gate: V3 style (sigmoid + bias, group-limited) and V2 style (softmax greedy), with exact ties
    s3 = rng.standard_normal(256).astype(np.float32)
    s3[10] = s3[200] = s3[37]  # ties: the lowest index must win
    b3 = (0.1 * rng.standard_normal(256)).astype(np.float32)
    b3[10] = b3[200] = b3[37]
    e, wt, sc = R.moe_gate(s3, b3, 8, True, 2.5, 1, 1, 8, 4)
    out["gate3_s"], out["gate3_b"], out["gate3_e"], out["gate3_w"], out["gate3_scores"] = s3, b3, e, wt, sc
    s2 = rng.standard_normal(64).astype(np.float32)
    s2[5] = s2[50]
    e, wt, sc = R.moe_gate(s2, None, 6, False, 1.0, 0, 0, 1, 1)
    out["gate2_s"], out["gate2_e"], out["gate2_w"], out["gate2_scores"] = s2, e, wt, sc
    # rope, both conventions
    v = rng.standard_normal(64).astype(np.float32)
    out["rope_in"] = v
    out["rope_v2"] = R.rope(v, 64, 1234, 10000.0, False)
    out["rope_v3"] = R.rope(v, 64, 1234, 10000.0, True)
    # attention
    H, hd, vd, T = 4, 192, 128, 70
    q = rng.standard_normal(H * hd).astype(np.float32)
    kb = rng.standard_normal((T, H * hd)).astype(np.float16).view(np.uint16)
    vb = rng.standard_normal((T, H * vd)).astype(np.float16).view(np.uint16)
    out["att_q"], out["att_k"], out["att_v"] = q, kb, vb
    out["att_y"] = R.attn_mha(q, kb, vb, H, hd, vd, T)
    lora, rope = 512, 64
    qc = (rng.standard_normal(H * lora) / 8).astype(np.float32)
    qr = rng.standard_normal(H * rope).astype(np.float32)
    ckv = rng.standard_normal((T, lora)).astype(np.float16).view(np.uint16)
    kr = rng.standard_normal((T, rope)).astype(np.float16).view(np.uint16)
    out["mla_qc"], out["mla_qr"], out["mla_ckv"], out["mla_kr"] = qc, qr, ckv, kr
    out["mla_y"] = R.attn_mla(qc, qr, ckv, kr, H, 192, lora, rope, T)
    # scalar codecs
    vals = np.concatenate([rng.standard_normal(200).astype(np.float32) * 10,
                           np.array([0, -0.0, 1e-8, 6e-8, 6.1e-5, 65504, 65519, 65520, 1e6, -1e6, 5.96e-8, 2.98e-8,
                                     2.9802322e-08], np.float32)])
    out["codec_in"] = vals
    out["codec_f16"] = np.array([R.lib.ref_float_to_half(float(v)) for v in vals], np.uint16)
    out["codec_f8"] = np.array([R.lib.ref_float_to_f8e5m2(float(v)) for v in vals], np.uint8)
    out["codec_h2f"] = np.array([R.lib.ref_half_to_float(int(h)) for h in range(0, 65536, 97)], np.float32)
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **out)
    print("ops ok")


def sampler_cases():
    """(logits, temperature, top_p, seed) of the sampler fixture: the logits are functions of the seed only."""
    V = synth.preset("tiny_v3", "fp16", False).vocab_size
    cases = []
    for k in range(48):
        rng = np.random.default_rng(1000 + k)
        spread = [0.5, 2.0, 6.0, 12.0][k % 4]  # flat ... peaked distributions
        logits = (rng.standard_normal(V) * spread).astype(np.float32)
        if k % 8 == 7:
            logits[rng.integers(0, V, 3)] = logits.max()  # ties of the maximum
        temperature = [1.0, 0.7, 1.5, 0.0][k % 4 if k % 12 else 3]
        top_p = [0.95, 1.0, 0.5, 0.9][(k // 4) % 4]
        cases.append((logits, np.float32(temperature), np.float32(top_p), 7 + k))
    return V, cases


def gen_sampler(R):
    """Sampler::sample / sample_argmax of the reference itself (oracle/ref_shim.cpp ref_sample) -> tests/golden/sampler.npz."""
    V, cases = sampler_cases()
    c = synth.preset("tiny_v3", "fp16", False)
    d = tempfile.mkdtemp(prefix="dsk_gold_sampler_")
    synth.write_dseek(d, c, synth.synth_model(c, seed=1))
    S = R.session(d, c)
    toks, coins = [], []
    for logits, t, p, seed in cases:
        tok, coin = S.sample(logits, float(t), float(p), int(seed))
        toks.append(tok)
        coins.append(coin)
    S.close()
    np.savez_compressed(os.path.join(GOLD, "sampler.npz"), token=np.array(toks, np.int32), coin=np.array(coins, np.float32),
                        temperature=np.array([c_[1] for c_ in cases], np.float32), top_p=np.array([c_[2] for c_ in cases], np.float32),
                        seed=np.array([c_[3] for c_ in cases], np.int32), logits_crc=np.array([int(np.sum(c_[0].view(np.uint32) % 65521)) for c_ in cases], np.int64))
    print("sampler ok", toks[:8])


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    orc.build()
    R = orc.Ref()
    R.set_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "sampler":
        gen_sampler(R)
        sys.exit(0)
    gen_ops(R)
    gen_models(R)
    gen_sampler(R)
