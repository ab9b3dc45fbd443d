"""Shared parity metrics and model-case helpers for the test-suite."""
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def rel_inf(a, b):
    """max|a-b| / max|b|: the 'relative' of BASELINE.json's "within 1e-3 relative for fp logits",
    taken against the scale of the whole vector (a per-element ratio is meaningless for logits
    that happen to be near zero)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def model_sha(T) -> str:
    h = hashlib.sha256()
    for name in sorted(T):
        h.update(name.encode())
        h.update(np.ascontiguousarray(T[name].data).tobytes())
        if T[name].scale is not None:
            h.update(np.ascontiguousarray(T[name].scale).tobytes())
    return h.hexdigest()


MODEL_CASES = [  # must match tools/make_golden.py
    ("tiny_v3", "q2_k", False, 7), ("tiny_v3", "q2_k", True, 7), ("tiny_v3", "q3_k", False, 8),
    ("tiny_v3", "q3_k", True, 8), ("tiny_v3", "f8e5m2", False, 9), ("tiny_v3", "f8e5m2", True, 9),
    ("tiny_v2lite", "q2_k", False, 10), ("tiny_v2lite", "f8e5m2", False, 11), ("tiny_v2lite", "fp16", False, 12),
    ("tiny_v2lite", "fp32", False, 13),
]


def case_id(case):
    preset, quant, mla, seed = case
    return f"{preset}-{quant}-{'mla' if mla else 'mha'}"


def load_case(case):
    """Regenerate the synthetic model of a golden case and load the reference's recorded outputs."""
    from tools import synth
    preset, quant, mla, seed = case
    c = synth.preset(preset, quant, mla)
    T = synth.synth_model(c, seed=seed)
    g = np.load(os.path.join(GOLD, f"model_{preset}_{quant}_{'mla' if mla else 'mha'}_s{seed}.npz"))
    sha_ok = bytes(g["sha"]).hex() == model_sha(T)
    return c, T, g, sha_ok


def is_kquant(quant):
    return quant in ("q2_k", "q3_k")


def model_parity_stats(model, c, g):
    """Run a model object (oracle / HIP binding: .forward(token,pos), .routing()) over a golden
    case: the recorded 8-token sequence, then 16 independent single-token trials at pos 0."""
    V = c.vocab_size
    seq_errs, same, n = [], 0, 0
    for pos, t in enumerate(g["tokens"]):
        lo = model.forward(int(t) % V, pos)
        seq_errs.append(rel_inf(lo, g["logits"][pos]))
        e, _ = model.routing()
        for l in range(c.n_layers):
            if g["route_e"][pos][l][0] >= 0:
                n += 1
                same += int(np.array_equal(e[l], g["route_e"][pos][l]))
    errs0, same0, n0 = [], 0, 0
    for i, t in enumerate(g["tokens0"]):
        lo = model.forward(int(t) % V, 0)
        errs0.append(rel_inf(lo, g["logits0"][i]))
        e, _ = model.routing()
        for l in range(c.n_layers):
            if g["route0_e"][i][l][0] >= 0:
                n0 += 1
                same0 += int(np.array_equal(e[l], g["route0_e"][i][l]))
    return dict(seq_errs=seq_errs, errs0=errs0, routes=(same, n), routes0=(same0, n0))


def assert_model_parity(st, kquant, what=""):
    """Acceptance (BASELINE.json: top-k expert indices identical, logits within 1e-3 relative).

    Float-weight models (fp32 / fp16 / f8e5m2): every token of the free-running sequence and every
    independent trial within 1e-3 of the logit scale, routing identical everywhere.

    W2A8 / W3A8 models are discontinuous in their activations: a last-bit difference in an RMSNorm output can flip an
    int8 rounding that sits on a tie (quantize_row_q8_K_ref, src/quant.cpp:616-653), and a free-running model carries
    the flip forward (KV cache included).  The EXACT statement for these models is the teacher-forced, flip-audited
    test (tests/teacher.py, tests/test_teacher_forced_gpu.py): every staging point's codes equal the reference's except
    proven ties, every stage within 2e-5 on identical codes, expert indices identical.  Here only a smoke remains:
    at least half of the independent pos-0 trials see no tie at all and then agree to float precision (2e-5), and
    nothing is grossly wrong anywhere (< 0.3 of the logit scale: a layout or indexing bug gives O(1)).
    """
    seq, e0 = st["seq_errs"], st["errs0"]
    if kquant:
        frac = float(np.mean(np.array(e0) < 2e-5))
        assert frac >= 0.5, (what, "independent trials without a rounding tie, within 2e-5", frac, e0)
        assert max(seq + e0) < 0.3, (what, seq, e0)
        # a weak floor on free-running routing agreement: ties move a router logit by ~1e-4 of its scale, so
        # most decisions survive; the exact statement is the teacher-forced audit
        (same0, n0), (same, n) = st["routes0"], st["routes"]
        assert n0 == 0 or same0 >= 0.5 * n0, (what, "routing agreement of the independent pos-0 trials", same0, n0)
        assert n == 0 or same >= 0.3 * n, (what, "routing agreement over the free-running sequence", same, n)
    else:
        assert max(seq + e0) < 1e-3, (what, seq, e0)
        assert st["routes"][0] == st["routes"][1] and st["routes0"][0] == st["routes0"][1], (what, st["routes"], st["routes0"])
