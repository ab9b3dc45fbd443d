"""GPU parity, model level: dsk_forward (the replacement of Model::forward) vs the reference's
recorded outputs (tests/golden/model_*.npz), the CPU oracle and -- where the prebuilt
oracle/_ref/libdskref.so loads -- the unmodified reference running live on the host.
"""
import os
import tempfile

import numpy as np
import pytest

from tests.util import MODEL_CASES, assert_model_parity, case_id, is_kquant, load_case, model_parity_stats, rel_inf
from tools import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", MODEL_CASES, ids=case_id)
def test_model_vs_reference_golden(ctx, case):
    import dsk
    c, T, g, sha_ok = load_case(case)
    if not sha_ok:
        pytest.skip("synthetic weights differ from the fixture's (numpy RNG change)")
    M = dsk.Model(ctx, c, T)
    st = model_parity_stats(M, c, g)
    M.close()
    assert_model_parity(st, is_kquant(c.quant), "HIP vs reference")


@pytest.mark.parametrize("case", MODEL_CASES, ids=case_id)
def test_model_vs_oracle_live(ctx, oracle, case):
    """Same statistics against the oracle on tokens the fixtures do not contain."""
    import dsk
    c, T, _, _ = load_case(case)
    M, O = dsk.Model(ctx, c, T), oracle.model(c, T)
    toks = [11, 400, 23, 999, 7]
    g = dict(tokens=np.array(toks), tokens0=np.array([3, 33, 333, 600, 41, 52, 63, 74]))
    g["logits"], g["route_e"], g["logits0"], g["route0_e"] = [], [], [], []
    for pos, t in enumerate(toks):
        g["logits"].append(O.forward(t % c.vocab_size, pos))
        g["route_e"].append(O.routing()[0])
    for t in g["tokens0"]:
        g["logits0"].append(O.forward(int(t) % c.vocab_size, 0))
        g["route0_e"].append(O.routing()[0])
    st = model_parity_stats(M, c, g)
    M.close()
    O.close()
    assert_model_parity(st, is_kquant(c.quant), "HIP vs oracle")


def test_trace_x_per_layer_float_model(ctx, oracle):
    """Residual stream after every block, fp16 weights (no W.A8 discontinuity): 1e-4 per layer."""
    import dsk
    c = synth.preset("tiny_v3", "fp16", False)
    T = synth.synth_model(c, seed=21)
    M, O = dsk.Model(ctx, c, T), oracle.model(c, T)
    M.set_trace(True)
    for pos, t in enumerate([4, 8, 15, 16, 23, 42]):
        lg, lo = M.forward(t, pos), O.forward(t, pos)
        for l in range(c.n_layers):
            assert rel_inf(M.trace_x(l), O.trace_x(l)) < 1e-4, (pos, l)
        assert rel_inf(lg, lo) < 1e-4
        assert np.array_equal(M.routing()[0], O.routing()[0])
    M.close()
    O.close()


@pytest.mark.parametrize("quant,mla", [("q2_k", False), ("q2_k", True), ("f8e5m2", False)])
def test_graph_replay_equals_eager_and_is_deterministic(ctx, quant, mla):
    import dsk
    c = synth.preset("tiny_v3", quant, mla)
    T = synth.synth_model(c, seed=5)
    A, B = dsk.Model(ctx, c, T), dsk.Model(ctx, c, T)
    B.set_graph(False)
    for pos, t in enumerate([9, 99, 999, 5, 6, 7]):
        la, lb = A.forward(t, pos), B.forward(t, pos)
        assert np.array_equal(la, lb), pos  # same kernels, same order => bit-identical
    assert np.array_equal(A.forward(9, 0), A.forward(9, 0))
    A.close()
    B.close()


def test_hydrate_mode_matches_full_forward(ctx):
    """HYDRATE_KV_CACHE must leave the caches exactly as OUTPUT_LOGITS does (src/infer.cpp:1284-1287)."""
    import dsk
    c = synth.preset("tiny_v3", "q2_k", True)
    T = synth.synth_model(c, seed=6)
    A, B = dsk.Model(ctx, c, T), dsk.Model(ctx, c, T)
    toks = [3, 1, 4, 1, 5, 9, 2, 6]
    for pos, t in enumerate(toks[:-1]):
        A.forward(t, pos)
        assert B.forward(t, pos, dsk.MODE_HYDRATE_KV_CACHE) is None
    assert np.array_equal(A.forward(toks[-1], len(toks) - 1), B.forward(toks[-1], len(toks) - 1))
    A.close()
    B.close()


@pytest.mark.parametrize("mla", [False, True], ids=["mha", "mla"])
@pytest.mark.parametrize("arch", ["DeepseekV3ForCausalLM", "DeepseekV2ForCausalLM"], ids=["rope_v3", "rope_v2"])
def test_kv_ring_and_sink_rotation(ctx, oracle, mla, arch):
    """StreamingLLM ring (src/infer.cpp:1271-1277) with the two attention sinks re-rotated in f16
    every step once pos >= W (src/infer.cpp:1004-1020, 1099-1110); W = 8 so 24 tokens wrap twice."""
    import dsk
    c = synth.preset("tiny_v3", "fp16", mla, arch=arch, rs_original_max_position_embeddings=8, max_seq_len=8)
    T = synth.synth_model(c, seed=31)
    M, O = dsk.Model(ctx, c, T), oracle.model(c, T)
    rng = np.random.default_rng(2)
    for pos in range(24):
        t = int(rng.integers(0, c.vocab_size))
        assert rel_inf(M.forward(t, pos), O.forward(t, pos)) < 1e-3, pos
    M.close()
    O.close()


def test_live_reference_on_this_host(ctx, ref):
    """HIP vs the unmodified reference executing on the host CPU (prebuilt oracle/_ref)."""
    import dsk
    for quant, mla in (("f8e5m2", False), ("q2_k", True)):
        c = synth.preset("tiny_v3", quant, mla)
        T = synth.synth_model(c, seed=77)
        d = tempfile.mkdtemp()
        synth.write_dseek(d, c, T)
        S, M = ref.session(d, c), dsk.Model(ctx, c, T)
        g = dict(tokens=np.array([2, 3, 5, 7]), tokens0=np.arange(10, 18))
        g["logits"], g["route_e"], g["logits0"], g["route0_e"] = [], [], [], []
        for pos, t in enumerate(g["tokens"]):
            g["logits"].append(S.forward(int(t), pos))
            g["route_e"].append(S.routing()[0])
        for t in g["tokens0"]:
            g["logits0"].append(S.forward(int(t), 0))
            g["route0_e"].append(S.routing()[0])
        st = model_parity_stats(M, c, g)
        M.close()
        S.close()
        assert_model_parity(st, is_kquant(quant), "HIP vs live reference")


def test_active_bytes_formula(ctx):
    """dsk_model_active_bytes = true-block-size algorithmic bytes."""
    import dsk
    c = synth.preset("tiny_v3", "q2_k", False)
    M = dsk.Model(ctx, c, synth.synth_model(c, seed=1))
    H, hd = c.n_heads, c.head_dim
    q = lambda r, n: r * n // 256 * 84
    attn = q(c.q_lora_rank, c.dim) + q(H * hd, c.q_lora_rank) + q(c.kv_lora_rank + c.qk_rope_head_dim, c.dim) + \
        q(H * (c.qk_nope_head_dim + c.v_head_dim), c.kv_lora_rank) + q(c.dim, H * c.v_head_dim)
    norms = 4 * (2 * c.dim + c.q_lora_rank + c.kv_lora_rank)
    dense = 3 * q(c.hidden_dim, c.dim)
    moe = c.n_active_routed * 3 * q(c.moe_intermediate_size, c.dim) + 3 * q(c.n_shared_experts * c.moe_intermediate_size, c.dim) + \
        4 * c.n_routed_experts * c.dim + 4 * c.n_routed_experts
    pos = 5
    kv = (pos + 1) * H * (hd + c.v_head_dim) * 2
    want = q(1, c.dim) + c.n_layers * (attn + norms + kv) + dense + 2 * moe + 4 * c.dim + q(c.vocab_size, c.dim)
    assert M.active_bytes(pos) == pytest.approx(want, rel=1e-12)
    M.close()


def test_error_paths_do_not_abort(ctx):
    """The reference assert(false)s (src/model.cpp:131-132); the boundary returns codes + messages."""
    import ctypes as C
    import dsk
    L = dsk.lib()
    c = synth.preset("tiny_v3", "q2_k", False)
    T = synth.synth_model(c, seed=1)
    M = dsk.Model(ctx, c, T)
    logits = np.zeros(c.vocab_size, np.float32)
    lp = logits.ctypes.data_as(dsk.c_f)
    assert L.dsk_forward(M.h, c.vocab_size, 0, 1, lp) == -1 and b"token" in L.dsk_last_error()
    assert L.dsk_forward(M.h, 0, -1, 1, lp) == -1
    assert L.dsk_forward(M.h, 0, 0, 7, lp) == -1
    assert L.dsk_forward(M.h, 0, 0, 1, None) == -1
    assert L.dsk_forward(M.h, 0, c.max_seq_len, 1, lp) == -1 and b"max_seq_len" in L.dsk_last_error()
    assert L.dsk_forward(None, 0, 0, 1, lp) == -1
    M.close()
    # forward before finalize / missing tensor / wrong shape / wrong byte count / double bind
    h = C.c_void_p()
    dc = dsk.make_config(c)
    assert L.dsk_model_create(ctx.h, C.byref(dc), C.byref(h)) == 0
    assert L.dsk_forward(h, 0, 0, 1, lp) == -4
    assert L.dsk_model_finalize(h) == -4 and b"not bound" in L.dsk_last_error()
    t = T["model.norm.weight"]
    bad_shape = synth.shape4((c.dim + 1,))
    assert L.dsk_model_bind(h, 1, -1, 0, bad_shape.ctypes.data_as(dsk.c_i32), t.data.ctypes.data, t.data.nbytes) == -1
    ok_shape = synth.shape4((c.dim,))
    assert L.dsk_model_bind(h, 1, -1, 0, ok_shape.ctypes.data_as(dsk.c_i32), t.data.ctypes.data, t.data.nbytes - 4) == -1
    assert L.dsk_model_bind(h, 1, -1, 3, ok_shape.ctypes.data_as(dsk.c_i32), t.data.ctypes.data, t.data.nbytes) == -1
    assert L.dsk_model_bind(h, 1, -1, 0, ok_shape.ctypes.data_as(dsk.c_i32), t.data.ctypes.data, t.data.nbytes) == 0
    assert L.dsk_model_bind(h, 1, -1, 0, ok_shape.ctypes.data_as(dsk.c_i32), t.data.ctypes.data, t.data.nbytes) == -4
    assert L.dsk_model_bind(h, 20, 0, 3, ok_shape.ctypes.data_as(dsk.c_i32), t.data.ctypes.data, t.data.nbytes) == -1  # WC in an MHA model
    assert L.dsk_model_destroy(h) == 0
    # configuration the reference rejects: MLA without q_lora_rank (src/infer.cpp:1057)
    bad = synth.preset("tiny_v2lite", "fp16", False)
    bad.use_mla = True
    dc = dsk.make_config(bad)
    assert L.dsk_model_create(ctx.h, C.byref(dc), C.byref(h)) == -1 and b"q_lora_rank" in L.dsk_last_error()


def test_synthesized_weights_are_valid_and_deterministic(ctx):
    import dsk
    for quant in ("q2_k", "q3_k", "f8e5m2", "fp16"):
        c = synth.preset("tiny_v3", quant, quant != "fp16")
        A, B = dsk.Model(ctx, c, None, synth_seed=123), dsk.Model(ctx, c, None, synth_seed=123)
        la, lb = A.forward(17, 0), B.forward(17, 0)
        assert np.all(np.isfinite(la)) and np.array_equal(la, lb) and np.std(la) > 0
        e, w = A.routing()
        moe = e[c.first_k_dense_replace:]
        assert moe.min() >= 0 and moe.max() < c.n_routed_experts and all(len(set(r)) == len(r) for r in moe)
        A.close()
        B.close()


@pytest.mark.parametrize("quant", ["q2_k", "f8e5m2"])
def test_expert_sharded_path_dry_run_on_one_gpu(quant):
    """The N > 1 compute path without N GPUs: two models on this GPU own shard 0 / 1 of 2 (dsk_comm_init
    with uid = NULL: no communicator, the all-reduce is skipped).  For the last MoE layer every routed
    slot must be computed by exactly its owner, bit-identical to the unsharded model, and be exactly 0
    elsewhere (so that the sum all-reduce is exact); the shared expert is replicated; the stand-alone
    combine kernel of the sharded path must add the local slots in k order."""
    import dsk
    c = synth.preset("tiny_v3", quant, False, n_layers=2, first_k_dense_replace=1)
    T = synth.synth_model(c, seed=9)
    K, E = c.n_active_routed, c.n_routed_experts
    per = -(-E // 2)
    ctxs = [dsk.Ctx(0) for _ in range(3)]
    ctxs[1].comm_init_dry(0, 2)
    ctxs[2].comm_init_dry(1, 2)
    Ms = [dsk.Model(x, c, T) for x in ctxs]
    for M in Ms:
        M.set_trace(True)
    for tok in (5, 77, 300):
        for M in Ms:
            M.forward(tok, 0)
        (eC, wC), (eA, _), (eB, _) = (M.routing() for M in Ms)
        assert np.array_equal(eC, eA) and np.array_equal(eC, eB)  # the router is replicated
        oC, oA, oB = (M.slot_outputs() for M in Ms)
        owners = eC[1] // per
        assert set(owners) <= {0, 1}
        for k in range(K):
            own, other = (oA, oB) if owners[k] == 0 else (oB, oA)
            assert np.array_equal(own[k], oC[k]), (tok, k)
            assert not np.any(other[k]), (tok, k)
        if c.n_shared_experts > 0:
            assert np.array_equal(oA[K], oC[K]) and np.array_equal(oB[K], oC[K])
        # combine kernel: x_C - x_shard = sum over the slots the shard does NOT own of w_k * out_k
        xC = Ms[0].trace_x(1)
        for M, rank in ((Ms[1], 0), (Ms[2], 1)):
            missing = sum((wC[1][k] * oC[k] for k in range(K) if owners[k] != rank), np.zeros(c.dim, np.float32))
            assert rel_inf(xC - M.trace_x(1), missing) < 1e-4 or np.abs(missing).max() == 0, (tok, rank)
    for M in Ms:
        M.close()
    for x in ctxs:
        x.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_expert_sharding_exchange_emulated_on_the_host(world):
    """World sizes 2 / 4 / 8 on ONE GPU (dry-run shards, no communicator): after the sharded W2 launch every routed slot
    must be non-zero on exactly the owner of its expert and zero everywhere else, so the sum all-reduce of the K x dim
    slot buffer the real path issues (forward.cpp ffn) is exact and order-independent: the host-side sum over the
    shards must equal the single-GPU slot outputs BIT for bit, at every MoE layer reached through the trace."""
    import dsk
    c = synth.preset("tiny_v3", "q2_k", False, n_layers=2, first_k_dense_replace=1)
    T = synth.synth_model(c, seed=19)
    K, E = c.n_active_routed, c.n_routed_experts
    full_ctx = dsk.Ctx(0)
    full = dsk.Model(full_ctx, c, T)
    ctxs, shards = [], []
    for r in range(world):
        x = dsk.Ctx(0)
        x.comm_init_dry(r, world)
        ctxs.append(x)
        shards.append(dsk.Model(x, c, T))
    lib = dsk.lib()
    for tok in (3, 500, 1001):
        full.forward(tok, 0)
        e_full = full.routing()[0][1]
        o_full = full.slot_outputs()
        outs = []
        for M in shards:
            M.forward(tok, 0)
            assert np.array_equal(M.routing()[0][1], e_full)  # routing is replicated, no communication needed
            outs.append(M.slot_outputs())
        for k in range(K):
            base, cnt = (dsk.C.c_int(), dsk.C.c_int())
            owners = []
            for r in range(world):
                dsk.check(lib.dsk_expert_shard(E, world, r, dsk.C.byref(base), dsk.C.byref(cnt)))
                if base.value <= e_full[k] < base.value + cnt.value:
                    owners.append(r)
            assert len(owners) == 1
            for r in range(world):
                if r != owners[0]:
                    assert not np.any(outs[r][k]), (world, tok, k, r)
            total = np.sum([o[k] for o in outs], axis=0, dtype=np.float32)
            assert np.array_equal(total, o_full[k]), (world, tok, k)
        for r in range(world):  # the shared expert is replicated
            assert np.array_equal(outs[r][K], o_full[K])
    for M in shards + [full]:
        M.close()
    for x in ctxs + [full_ctx]:
        x.close()


def test_bench_dry_shard_mode_runs():
    """bench.py's sharded code path (rank r of w, no communicator) on a small model: the launcher logic of --gpus N minus
    the RCCL exchange, which needs N GPUs (never available to the builder: stated in README.md)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--model", "tiny_v3", "--steps", "4", "--warmup", "2", "--ctx", "64",
                        "--no-cpu-baseline", "--no-extras", "--dry-shard", "1/4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["value"] > 0 and "dry run of expert shard 1/4" in d["config"]["parallelism"]
    assert "DeepSeek-V3 Q2_K" not in d["metric"]  # a small model is not labelled with the headline metric


def test_forward_argmax_matches_host_argmax(ctx):
    """dsk_forward_argmax = dsk_forward + Sampler::sample_argmax (first maximum), graph and eager."""
    import dsk
    for quant, mla in (("q2_k", False), ("fp16", True)):
        c = synth.preset("tiny_v3", quant, mla)
        T = synth.synth_model(c, seed=13)
        A, B = dsk.Model(ctx, c, T), dsk.Model(ctx, c, T)
        tok_a = tok_b = 17
        for pos in range(6):  # first call eager, later ones replay the captured graph
            lg = A.forward(tok_a, pos)
            nxt = B.forward_argmax(tok_b, pos)
            assert nxt == int(np.argmax(lg)), pos  # np.argmax also returns the first maximum
            tok_a = tok_b = nxt
        A.close()
        B.close()


@pytest.mark.timeout(600)
def test_mla_long_context_regime_matches_oracle(ctx, oracle):
    """kv_len >= 320 (MLA_FLASH_MIN_KV) switches the MLA path to the matrix-core attention (mla_flash_kernel partials
    merged in mla_head_kernel, its own captured graph; positions per chunk follow the context length).  Float weights
    (no W.A8 discontinuity): logits within 1e-3 of the oracle on both sides of the switch and where the chunk length
    steps from 32 to 64 positions (kv_len 2048 is out of a tiny model's reach: 704 / 64 = 11 -> one 32-block; the
    two-block case is covered at op level), routing identical."""
    import dsk
    c = synth.preset("tiny_v3", "fp16", True, kv_lora_rank=512, qk_rope_head_dim=64, max_seq_len=832)
    T = synth.synth_model(c, seed=31)
    M, O = dsk.Model(ctx, c, T), oracle.model(c, T)
    rng = np.random.default_rng(0)
    toks = rng.integers(0, c.vocab_size, 800)
    worst = 0.0
    for pos, t in enumerate(toks):
        if not (308 <= pos < 334 or pos >= 776):  # fill both caches cheaply
            M.forward(int(t), pos, mode=0)
            O.forward(int(t), pos, mode=0)
            continue
        lg, lo = M.forward(int(t), pos), O.forward(int(t), pos)
        worst = max(worst, rel_inf(lg, lo))
        assert np.array_equal(M.routing()[0], O.routing()[0]), pos
    M.close()
    O.close()
    assert worst < 1e-3, worst


@pytest.mark.timeout(900)
def test_mha_long_context_regime_matches_oracle(ctx, oracle, monkeypatch):
    """From kv_len >= the split threshold on (default 1024; 256 here) the MHA path runs 256 / n_heads workgroups per
    head, each over a share of the cached positions, merged by the head's last split (head_attn_kernel), in its own
    captured graph.  Float weights: logits within 1e-3 of the oracle on both sides of the switch, routing identical;
    the ring wraps inside the long regime (attention sinks are rotated by split 0 only)."""
    import dsk
    c = synth.preset("tiny_v3", "fp16", False, max_seq_len=320)
    c.rs_original_max_position_embeddings = 288  # ring of 288 positions: wraps at pos 288
    T = synth.synth_model(c, seed=33)
    M, O = dsk.Model(ctx, c, T, options={"mha_split_min": 256}), oracle.model(c, T)
    rng = np.random.default_rng(1)
    toks = rng.integers(0, c.vocab_size, 310)
    worst = 0.0
    for pos, t in enumerate(toks):
        if pos < 248 and pos % 16:  # fill both caches cheaply, check now and then
            M.forward(int(t), pos, mode=0)
            O.forward(int(t), pos, mode=0)
            continue
        lg, lo = M.forward(int(t), pos), O.forward(int(t), pos)
        worst = max(worst, rel_inf(lg, lo))
        assert np.array_equal(M.routing()[0], O.routing()[0]), pos
    M.close()
    O.close()
    assert worst < 1e-3, worst


LOADER_CASES = [("tiny_v3", "q2_k", False), ("tiny_v3", "q3_k", True), ("tiny_v3", "f8e5m2", False), ("tiny_v3", "fp16", True),
                ("tiny_v2lite", "q2_k", False), ("tiny_v2lite", "fp32", False)]


@pytest.mark.parametrize("preset,quant,mla", LOADER_CASES, ids=[f"{a}-{b}-{'mla' if c else 'mha'}" for a, b, c in LOADER_CASES])
def test_dseek_loader_gives_the_same_model_as_the_bind_walk(ctx, tmp_path, preset, quant, mla):
    """dsk_model_load_dseek (file ranges -> pinned staging ring -> HBM, csrc/loader.cpp) against the same tensors bound
    one by one from host memory (dsk_model_bind, the walk of INTEGRATION.md): bit-identical logits and routing, across
    three shard files."""
    import dsk
    c = synth.preset(preset, quant, mla)
    T = synth.synth_model(c, seed=21)
    d = str(tmp_path / "ckpt")
    synth.write_dseek(d, c, T, shards=3)
    A, B = dsk.Model(ctx, c, T), dsk.Model.from_dseek(ctx, d)
    st = B.load_stats
    n_scale = sum(1 for t in T.values() if t.scale is not None)
    assert st.n_files == 3 and st.n_tensors == len(T) + n_scale
    assert st.file_bytes == st.staged_bytes == sum(t.data.nbytes + (t.scale.nbytes if t.scale is not None else 0) for t in T.values())
    assert st.seconds > 0
    for pos, tok in enumerate([3, 200, 41, 7]):
        assert np.array_equal(A.forward(tok, pos), B.forward(tok, pos)), pos
        assert np.array_equal(A.routing()[0], B.routing()[0])
    A.close()
    B.close()


def test_dseek_loader_on_a_sharded_context_reads_only_its_experts(tmp_path):
    """Expert-sharded rank (dry run, no communicator): the loader stages only the experts this rank owns, and the model
    equals the one bound from memory on the same shard."""
    import dsk
    c = synth.preset("tiny_v3", "q2_k", False, n_layers=2, first_k_dense_replace=1)
    T = synth.synth_model(c, seed=22)
    d = str(tmp_path / "ckpt")
    synth.write_dseek(d, c, T, shards=2)
    routed = sum(t.data.nbytes for n, t in T.items() if ".mlp.w" in n and t.data.ndim == 3)
    for rank in (0, 1):
        x1, x2 = dsk.Ctx(0), dsk.Ctx(0)
        x1.comm_init_dry(rank, 2)
        x2.comm_init_dry(rank, 2)
        A, B = dsk.Model(x1, c, T), dsk.Model.from_dseek(x2, d)
        st = B.load_stats
        assert st.file_bytes - st.staged_bytes == routed // 2, (st.file_bytes, st.staged_bytes, routed)
        for pos, tok in enumerate([9, 310]):
            assert np.array_equal(A.forward(tok, pos), B.forward(tok, pos))
            assert np.array_equal(A.slot_outputs(), B.slot_outputs())
        A.close()
        B.close()
        x1.close()
        x2.close()


def test_dseek_loader_errors(ctx, tmp_path):
    import dsk
    c = synth.preset("tiny_v3", "q2_k", False)
    T = synth.synth_model(c, seed=23)
    # a tensor the constructors require is missing
    T2 = dict(T)
    del T2["model.layers.0.attn.wo.weight"]
    synth.write_dseek(str(tmp_path / "a"), c, T2)
    with pytest.raises(dsk.DskError, match="attn.wo.weight is missing"):
        dsk.Model.from_dseek(ctx, str(tmp_path / "a"))
    # wrong byte count for the configuration (a different hidden size in the metadata)
    c3 = synth.preset("tiny_v3", "q2_k", False, hidden_dim=c.hidden_dim * 2)
    synth.write_dseek(str(tmp_path / "b"), c3, T)
    with pytest.raises(dsk.DskError, match="bytes"):
        dsk.Model.from_dseek(ctx, str(tmp_path / "b"))
    # the context survives a failed load
    synth.write_dseek(str(tmp_path / "c"), c, T)
    M = dsk.Model.from_dseek(ctx, str(tmp_path / "c"))
    assert np.all(np.isfinite(M.forward(5, 0)))
    M.close()


def test_forward_sample_is_forward_plus_sampler(ctx, oracle):
    """dsk_forward_sample = dsk_forward + Sampler::sample (src/sampler.cpp:41-75) with the caller's random draw:
    the same token as the op on the model's own logits (eager, then captured graph; parameters change per token),
    the oracle's token on those logits, and temperature 0 is the argmax step."""
    import dsk
    c = synth.preset("tiny_v3", "q2_k", False)
    T = synth.synth_model(c, seed=14)
    A, B = dsk.Model(ctx, c, T), dsk.Model(ctx, c, T)
    rng = np.random.default_rng(2)
    tok = 17
    for pos in range(8):
        t = [1.0, 0.7, 0.0, 1.3][pos % 4]
        p = [0.95, 1.0, 0.9, 0.5][pos % 4]
        coin = float(np.float32(rng.uniform()))
        lg = A.forward(tok, pos)
        nxt = B.forward_sample(tok, pos, t, p, coin)
        assert nxt == ctx.sample(lg, t, p, coin), pos
        want = oracle.sample(lg, t, p, coin)
        assert nxt == want or t > 0, (pos, nxt, want)  # (near-ties: tests/test_ops_gpu.py)
        tok = nxt
    A.close()
    B.close()


def test_forward_prob_is_the_softmax_probability_of_the_index(ctx, oracle):
    """dsk_forward_prob = dsk_forward + Sampler::sample_prob (src/sampler.cpp:12-26), the per-token term of
    run_perplexity; float path: 1e-4 relative to the f64 softmax of the model's own logits (the reference's left-to-right
    f32 sum is itself ~1e-5 away from it); interleaved with sampling steps on the same model (shared parameters)."""
    import dsk
    c = synth.preset("tiny_v3", "fp16", False)
    T = synth.synth_model(c, seed=15)
    A, B = dsk.Model(ctx, c, T), dsk.Model(ctx, c, T)
    toks = [3, 99, 512, 7, 1000, 64]
    for pos in range(len(toks) - 1):
        lg = A.forward(toks[pos], pos).astype(np.float64)
        pr = np.exp(lg - lg.max())
        pr /= pr.sum()
        got = B.forward_prob(toks[pos], pos, toks[pos + 1])
        assert abs(got - pr[toks[pos + 1]]) <= 1e-4 * pr[toks[pos + 1]] + 1e-12, pos
        want = oracle.sample_prob(lg.astype(np.float32), toks[pos + 1])  # the reference's left-to-right f32 sums
        assert abs(got - want) <= 1e-4 * want + 1e-12, pos
        if pos == 2:  # a sampling step in between must not disturb the probability mode (and vice versa)
            assert B.forward_sample(toks[pos], pos, 1.0, 0.95, 0.5) == ctx.sample(lg.astype(np.float32), 1.0, 0.95, 0.5)
    with pytest.raises(dsk.DskError):
        B.forward_prob(3, 0, c.vocab_size)
    A.close()
    B.close()


@pytest.mark.parametrize("quant,mla", [("fp16", False), ("f8e5m2", True)], ids=["fp16-mha", "f8e5m2-mla"])
def test_dseek_loader_vs_the_reference_reading_the_same_files(ctx, ref, tmp_path, quant, mla):
    """The same checkpoint directory read by the reference (YALMData + Model, oracle/_ref) and by dsk_model_load_dseek:
    float paths, logits within 1e-3, identical routing, token by token."""
    import dsk
    c = synth.preset("tiny_v3", quant, mla)
    d = str(tmp_path / "ckpt")
    synth.write_dseek(d, c, synth.synth_model(c, seed=24), shards=2)
    M, S = dsk.Model.from_dseek(ctx, d), ref.session(d, c)
    for pos, tok in enumerate([11, 250, 3, 77, 512]):
        lh, lr = M.forward(tok, pos), S.forward(tok, pos)
        assert rel_inf(lh, lr) < 1e-3, pos
        assert np.array_equal(M.routing()[0], S.routing()[0]), pos
    M.close()
    S.close()


@pytest.mark.parametrize("quant,mla", [("q2_k", True), ("q3_k", False)])
def test_repacked_plane_layout_checkpoint_loads_bit_identically(ctx, quant, mla):
    """tools/repack.py persists the engine's plane layout; dsk_model_load_dseek copies the planes straight
    into HBM (no repack kernels).  Same device bytes => bit-identical logits, routing and slot outputs; an expert-sharded
    dry run reads only its own experts' ranges of every plane."""
    import dsk
    from tools import repack
    c = synth.preset("tiny_v3", quant, mla)
    T = synth.synth_model(c, seed=29)
    d, d2 = tempfile.mkdtemp(prefix="dsk_ref_"), tempfile.mkdtemp(prefix="dsk_planes_")
    try:
        synth.write_dseek(d, c, T, shards=2, tokenizer=True)
        repack.repack(d, d2)
        A, B = dsk.Model.from_dseek(ctx, d), dsk.Model.from_dseek(ctx, d2)
        assert B.load_stats.n_tensors == A.load_stats.n_tensors
        for pos, tok in enumerate([4, 90, 1000, 17]):
            assert np.array_equal(A.forward(tok, pos), B.forward(tok, pos)), pos
            assert np.array_equal(A.routing()[0], B.routing()[0])
            assert np.array_equal(A.slot_outputs(), B.slot_outputs())
        A.close()
        B.close()
        x = dsk.Ctx(0)
        x.comm_init_dry(1, 2)
        S1, S2 = dsk.Model.from_dseek(x, d), dsk.Model.from_dseek(x, d2)
        assert S2.load_stats.staged_bytes == S1.load_stats.staged_bytes < B.load_stats.staged_bytes
        S1.forward(5, 0)
        S2.forward(5, 0)
        assert np.array_equal(S1.slot_outputs(), S2.slot_outputs())
        S1.close()
        S2.close()
        x.close()
    finally:
        for dd in (d, d2):
            for f in os.listdir(dd):
                os.unlink(os.path.join(dd, f))
            os.rmdir(dd)
