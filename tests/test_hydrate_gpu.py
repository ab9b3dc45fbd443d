"""Batched prompt ingestion: dsk_hydrate against its own definition.

The reference feeds a prompt to Model::forward one token at a time (src/main.cpp:312-319, HYDRATE_KV_CACHE for all tokens but
the last; src/infer.cpp:1284-1287).  dsk_hydrate is defined as that loop over dsk_forward and runs it as batched launches
(kernels_hydrate.hip: every weight matrix read once per chunk, the Q2_K row products as i8 GEMMs on the matrix pipe) when the
model qualifies.  The bar is BIT-IDENTITY with the loop on the same model: every KV-cache row the prompt wrote, the residual
stream after every block for every token, and the last token's logits - which covers Q8_K codes, integer sums and routing
(a single different code or expert changes the bits downstream).  Parity of the loop itself with the reference is the
business of the other test files; this one proves that the batched path is the same function.
"""
import numpy as np
import pytest

from tools import synth

pytestmark = pytest.mark.gpu


def _caches(M, c, n_rows):
    H, hd, vd = c.n_heads, c.qk_nope_head_dim + c.qk_rope_head_dim, c.v_head_dim
    out = []
    for l in range(c.n_layers):
        if c.use_mla:
            out.append((M.get_cache_rows(l, "nope_cache", 0, n_rows, c.kv_lora_rank), M.get_cache_rows(l, "rope_cache", 0, n_rows, c.qk_rope_head_dim)))
        else:
            out.append((M.get_cache_rows(l, "k_cache", 0, n_rows, H * hd), M.get_cache_rows(l, "v_cache", 0, n_rows, H * vd)))
    return out


def _loop(M, tokens, pos0, trace=False):
    """the definition: one dsk_forward per token; returns the last logits (+ x after every block per token)"""
    import dsk
    xs = []
    lo = None
    for i, t in enumerate(tokens):
        last = i == len(tokens) - 1
        lo = M.forward(int(t), pos0 + i, dsk.MODE_OUTPUT_LOGITS if last else dsk.MODE_HYDRATE_KV_CACHE)
        if trace:
            xs.append(np.stack([M.trace_x(l) for l in range(M.cfg.n_layers)]))
    return lo, (np.stack(xs) if trace else None)


def compare_hydrate(ctx, c, T=None, seed=None, tokens=(), pos0=0, chunk=None, report=None):
    """two models of the same weights: A runs the per-token loop, B dsk_hydrate; returns a dict of mismatch counts"""
    import dsk
    opts = {"q2k_tiles": 2}
    A = dsk.Model(ctx, c, T, synth_seed=seed, options=opts)
    ob = dict(opts)
    if chunk:
        ob["hydrate_chunk"] = chunk
    B = dsk.Model(ctx, c, T, synth_seed=seed, options=ob)
    try:
        assert B.hydrate_why_not() == "", B.hydrate_why_not()
        A.set_trace(True)
        B.set_trace(True)
        tokens = [int(t) % c.vocab_size for t in tokens]
        pre = [(7 * i + 3) % c.vocab_size for i in range(pos0)]
        if pos0:  # an existing context, written by the loop on A and by an earlier dsk_hydrate call on B
            _loop(A, pre, 0)
            B.hydrate(pre, 0, dsk.MODE_HYDRATE_KV_CACHE)
        la, xa = _loop(A, tokens, pos0, trace=True)
        lb = B.hydrate(tokens, pos0, dsk.MODE_OUTPUT_LOGITS)
        P = len(tokens)
        res = {"batched": B.info("hydrate_batched_tokens"), "looped": B.info("hydrate_looped_tokens")}
        cap = chunk or 512  # (the engine's default chunk)
        first_of_last_chunk = ((P - 1) // cap) * cap
        bad_x = []
        for i in range(first_of_last_chunk, P):  # the batched trace holds the last chunk
            for l in range(c.n_layers):
                xb = B.hydrate_trace_x(l, i - first_of_last_chunk)
                if not np.array_equal(xb, xa[i][l]):
                    bad_x.append((i, l, float(np.abs(xb - xa[i][l]).max() / max(1e-30, np.abs(xa[i][l]).max()))))
        res["bad_x"] = bad_x
        ca, cb = _caches(A, c, pos0 + P), _caches(B, c, pos0 + P)
        bad_kv = []
        for l in range(c.n_layers):
            for name, a, b in (("k", ca[l][0], cb[l][0]), ("v", ca[l][1], cb[l][1])):
                rows = np.nonzero((a != b).any(axis=1))[0]
                if rows.size:
                    bad_kv.append((l, name, rows[:8].tolist(), int(rows.size)))
        res["bad_kv"] = bad_kv
        res["logits_equal"] = bool(np.array_equal(la, lb))
        res["logits_err"] = float(np.abs(la - lb).max() / max(1e-30, np.abs(la).max()))
        if report is not None:
            report(res)
        return res
    finally:
        A.close()
        B.close()


def _assert_identical(res, P, pos0):
    assert res["batched"] == P + pos0 and res["looped"] == 0, res
    assert not res["bad_x"], res["bad_x"][:6]
    assert not res["bad_kv"], res["bad_kv"][:6]
    assert res["logits_equal"], res["logits_err"]


@pytest.mark.parametrize("mla", [False, True], ids=["mha", "mla"])
@pytest.mark.parametrize("P,pos0,chunk", [(1, 0, None), (2, 0, None), (5, 3, None), (13, 2, None), (16, 0, None), (17, 0, None), (37, 5, None), (37, 0, 16), (48, 9, 7)])
def test_hydrate_equals_the_loop_tiny(ctx, P, pos0, chunk, mla):
    """tiny DeepSeek-V3 Q2_K model, MHA and MLA (1 dense + 2 MoE blocks, 16 experts top-4, rows of <= 8 blocks: single-block items):
    token quads that are full, ragged and single; 13 - 48 tokens: the 16-token GEMM forms (plain matrices from 12 tokens on, expert
    tasks from 6 rows on) with ragged last chunks; chunks that split the prompt (attention over rows an earlier chunk wrote)"""
    c = synth.preset("tiny_v3", "q2_k", mla)
    T = synth.synth_model(c, seed=41)
    rng = np.random.default_rng(P * 31 + pos0)
    tokens = rng.integers(0, c.vocab_size, P)
    _assert_identical(compare_hydrate(ctx, c, T, tokens=tokens, pos0=pos0, chunk=chunk), P, pos0)


@pytest.mark.parametrize("mla", [False, True], ids=["mha", "mla"])
@pytest.mark.parametrize("P,pos0", [(1, 0), (16, 0), (37, 2), (128, 0)])
def test_hydrate_equals_the_loop_v3_width(ctx, P, pos0, mla):
    """1 dense + 1 MoE block at DeepSeek-V3 width (dim 7168: 4-block items with a full last item; wo's 64-block rows; 2048-wide
    hidden vectors: single-block items), 256 routed experts top-8 in 8 groups, 128 heads, MHA and MLA (absorbed weights: wc of
    65536 rows, per-head wv_b), weights synthesised in HBM"""
    c = synth.preset("v3", "q2_k", mla, n_layers=2, first_k_dense_replace=1, max_seq_len=192)
    rng = np.random.default_rng(P)
    tokens = rng.integers(0, c.vocab_size, P)
    _assert_identical(compare_hydrate(ctx, c, None, seed=11, tokens=tokens, pos0=pos0), P, pos0)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("P", [256, 512])
def test_hydrate_equals_the_loop_v3_width_one_bench_sized_chunk(ctx, P):
    """the configuration bench.py times: ONE chunk of 256 / 512 tokens at DeepSeek-V3 width (the engine's default chunk is 512; the
    hottest expert of such a chunk carries hundreds of rows: many 16-row passes per task, every plain matrix at 16 / 32 token chunks)"""
    c = synth.preset("v3", "q2_k", False, n_layers=2, first_k_dense_replace=1, max_seq_len=P + 8)
    tokens = np.random.default_rng(P).integers(0, c.vocab_size, P)
    _assert_identical(compare_hydrate(ctx, c, None, seed=17, tokens=tokens, pos0=0), P, 0)


@pytest.mark.timeout(900)
def test_mha_batches_through_the_split_context_regime(ctx):
    """MHA at DeepSeek-V3 width: from mha_split_min (1024) cached positions on decode runs two workgroups per head over halves of the
    context and merges un-normalised partials (head_attn_kernel) - another float association than a whole-context softmax (ADVICE
    r5: the batched path used to run one softmax there).  hyd_attn_kernel walks the same pieces and merges them with the same
    statements: KV rows and logits are the loop's across position 1023, every token batched"""
    import dsk
    c = synth.preset("v3", "q2_k", False, n_layers=2, first_k_dense_replace=1, max_seq_len=1056)
    pre = [(5 * i + 1) % c.vocab_size for i in range(1000)]
    tokens = [(17 * i + 3) % c.vocab_size for i in range(40)]
    A = dsk.Model(ctx, c, None, synth_seed=19, options={"q2k_tiles": 2})
    B = dsk.Model(ctx, c, None, synth_seed=19, options={"q2k_tiles": 2})
    assert B.hydrate_why_not() == ""
    _loop(A, pre, 0)
    B.hydrate(pre, 0, dsk.MODE_HYDRATE_KV_CACHE)
    assert B.info("hydrate_batched_tokens") == 1000
    la, _ = _loop(A, tokens, 1000)
    lb = B.hydrate(tokens, 1000, dsk.MODE_OUTPUT_LOGITS)
    assert B.info("hydrate_batched_tokens") == 1040 and B.info("hydrate_looped_tokens") == 0
    assert np.array_equal(la, lb)
    for (ka, va), (kb, vb) in zip(_caches(A, c, 1040), _caches(B, c, 1040)):
        assert np.array_equal(ka, kb) and np.array_equal(va, vb)
    A.close()
    B.close()


@pytest.mark.parametrize("level", [2, 1], ids=["tiles-everywhere", "default-layout"])
@pytest.mark.parametrize("mla", [False, True], ids=["mha", "mla"])
def test_hydrate_blocks_audited_on_the_oracle(ctx, oracle, mla, level):
    """Not 'batched == own loop' but 'batched == the reference': the golden tiny DeepSeek-V3 Q2_K models (tests/golden, seed 7), a
    37-token prompt through dsk_hydrate, and EVERY block audited on the oracle for ten of the tokens (tests/teacher.py
    HydrateDevice + BlockAuditor: Q8_K codes equal to the oracle's except proven ties, every GEMV / float stage on the device's codes
    within 2e-5 / 1e-4, this position's K / V (latent) cache row against the oracle's arithmetic to the last f16 place, attention over
    the rows the same chunk wrote, route_e identical on the device's router logits).  The classifier row of the call (the last token's
    logits) goes through the head audit's arithmetic as well.  `default-layout`: the engine's default options (planes for decode, tile
    copies for the batched path - hydrate.cpp hyd_tile_copies) under the same audit."""
    import dsk
    from tests import teacher
    c = synth.preset("tiny_v3", "q2_k", mla)
    T = synth.synth_model(c, seed=7)
    emb = T["model.embed.weight"]
    tokens = [int(t) for t in np.random.default_rng(37).integers(0, c.vocab_size, 37)]
    aud = teacher.BlockAuditor(oracle, c, T)
    worst, flips, logits = 0.0, 0, None
    for l in range(c.n_layers):
        M = dsk.Model(ctx, c, T, options={"q2k_tiles": level, "hydrate_tap_layer": l})
        M.set_trace(True)
        lg = M.hydrate(tokens, 0, dsk.MODE_OUTPUT_LOGITS)
        assert M.info("hydrate_batched_tokens") == len(tokens) and M.info("hydrate_looped_tokens") == 0, M.hydrate_why_not()
        assert logits is None or np.array_equal(lg, logits)  # a tap changes nothing
        logits = lg
        for i in list(range(0, 37, 4)) + [36]:
            x_in = oracle.embed_row(emb.quant, emb.data, c.dim, tokens[i]) if l == 0 else M.hydrate_trace_x(l - 1, i)
            A, x_out = aud.run(teacher.HydrateDevice(M, c, l, i), l, x_in, i)
            flips += A.total_flips()
            worst = max(worst, max(A.errs.values()))
        if l == c.n_layers - 1:  # final norm + classifier of the last token on the chunk's own stream (src/infer.cpp:1292-1316)
            Ah, lg_head = teacher.audit_head(oracle, c, T, M, M.hydrate_trace_x(l, 36))
            assert np.array_equal(lg_head, lg)  # the call's logits ARE the audited head's on that stream
            flips += Ah.total_flips()
            worst = max(worst, max(Ah.errs.values()))
        M.close()
    print(f"\n[batched prompt, tiny_v3 q2_k {'mla' if mla else 'mha'}, q2k_tiles={level}] {c.n_layers} blocks x 11 tokens: worst stage error {worst:.2e}, {flips} proven int8 ties")
    assert worst < teacher.FLOAT_TOL


def test_hydrate_across_the_ring_wrap_takes_the_loop_there(ctx):
    """positions at and past rs_original_max_position_embeddings rotate the sink keys in place, token by token
    (src/infer.cpp:1008-1020): dsk_hydrate batches up to the wrap and loops from there - still the same bits as the loop"""
    import dsk
    c = synth.preset("tiny_v3", "q2_k", False, rs_original_max_position_embeddings=24, max_seq_len=64)
    T = synth.synth_model(c, seed=43)
    tokens = [(11 * i + 5) % c.vocab_size for i in range(40)]
    A = dsk.Model(ctx, c, T, options={"q2k_tiles": 2})
    B = dsk.Model(ctx, c, T, options={"q2k_tiles": 2, "hydrate_chunk": 16})
    la, _ = _loop(A, tokens, 0)
    lb = B.hydrate(tokens, 0, dsk.MODE_OUTPUT_LOGITS)
    assert B.info("hydrate_batched_tokens") == 24 and B.info("hydrate_looped_tokens") == 16
    assert np.array_equal(la, lb)
    for (ka, va), (kb, vb) in zip(_caches(A, c, 24), _caches(B, c, 24)):
        assert np.array_equal(ka, kb) and np.array_equal(va, vb)
    A.close()
    B.close()


@pytest.mark.timeout(900)
def test_mla_batches_through_the_matrix_core_regime(ctx):
    """MLA at DeepSeek-V3 width: from mla_flash_min_kv (320) cached positions on decode scores on the matrix cores (mla_flash_kernel:
    chunk partials with an online softmax, merged per head by mla_head_kernel) - another association than the short-context
    kernel's.  Round 5 sent such positions to the loop; since round 6 the batched path launches decode's own flash kernel over the
    chunk's long-context tokens (token = third grid dimension, 16 tokens' partials at a time) and merges with the one body both
    paths share (attn_device.h mla_merge_partials).  Pinned here: a 300-token prompt (short regime only), then 90 tokens whose chunk
    STRADDLES position 319 (19 short-context tokens + 71 long ones: both kernels in one chunk, more than four flash launches), then
    a chunk deep inside the regime (positions 700-739 after a jump of the cache contents) - logits and every latent / rope cache row
    equal to the loop's bit for bit, nothing looped."""
    import dsk
    c = synth.preset("v3", "q2_k", True, n_layers=2, first_k_dense_replace=1, max_seq_len=768)
    pre = [(5 * i + 1) % c.vocab_size for i in range(300)]
    tokens = [(17 * i + 3) % c.vocab_size for i in range(90)]
    A = dsk.Model(ctx, c, None, synth_seed=13, options={"q2k_tiles": 2})
    B = dsk.Model(ctx, c, None, synth_seed=13, options={"q2k_tiles": 2})
    assert B.hydrate_why_not() == ""
    _loop(A, pre, 0)
    B.hydrate(pre, 0, dsk.MODE_HYDRATE_KV_CACHE)
    assert B.info("hydrate_batched_tokens") == 300
    la, _ = _loop(A, tokens, 300)
    lb = B.hydrate(tokens, 300, dsk.MODE_OUTPUT_LOGITS)
    assert B.info("hydrate_batched_tokens") == 390 and B.info("hydrate_looped_tokens") == 0
    assert np.array_equal(la, lb)
    for (ka, va), (kb, vb) in zip(_caches(A, c, 390), _caches(B, c, 390)):
        assert np.array_equal(ka, kb) and np.array_equal(va, vb)
    # deep inside the regime: rows 390-699 of both models filled with the same random f16 rows, then 40 more tokens
    rng = np.random.default_rng(3)
    for l in range(c.n_layers):
        for name, width in (("nope_cache", c.kv_lora_rank), ("rope_cache", c.qk_rope_head_dim)):
            rows = (0.5 * rng.standard_normal((310, width))).astype(np.float16).view(np.uint16)
            A.set_cache_rows(l, name, 390, rows)
            B.set_cache_rows(l, name, 390, rows)
    more = [(29 * i + 11) % c.vocab_size for i in range(40)]
    la, _ = _loop(A, more, 700)
    lb = B.hydrate(more, 700, dsk.MODE_OUTPUT_LOGITS)
    assert B.info("hydrate_batched_tokens") == 430 and B.info("hydrate_looped_tokens") == 0
    assert np.array_equal(la, lb)
    for (ka, va), (kb, vb) in zip(_caches(A, c, 740), _caches(B, c, 740)):
        assert np.array_equal(ka, kb) and np.array_equal(va, vb)
    A.close()
    B.close()


@pytest.mark.parametrize("quant,mla,level,copies", [("f8e5m2", False, 2, 1), ("q2_k", True, 0, 1), ("q2_k", False, 0, 1), ("q2_k", True, 1, 0),
                                                    ("q2_k", False, 1, 0), ("q3_k", False, 2, 1)])
def test_models_that_do_not_qualify_run_the_loop(ctx, quant, mla, level, copies):
    """float weights, planes everywhere (the expert stacks are never copied), the default layout with the tile copies switched off,
    Q3_K: dsk_hydrate IS the loop there (and says why)"""
    import dsk
    c = synth.preset("tiny_v3", quant, mla)
    T = synth.synth_model(c, seed=47)
    tokens = [3, 99, 512, 7, 1000, 64]
    A = dsk.Model(ctx, c, T, options={"q2k_tiles": level})
    B = dsk.Model(ctx, c, T, options={"q2k_tiles": level, "hydrate_tile_copies": copies})
    assert B.hydrate_why_not() != ""
    la, _ = _loop(A, tokens, 0)
    lb = B.hydrate(tokens, 0, dsk.MODE_OUTPUT_LOGITS)
    assert B.info("hydrate_batched_tokens") == 0 and B.info("hydrate_looped_tokens") == len(tokens)
    assert np.array_equal(la, lb)
    A.close()
    B.close()


@pytest.mark.parametrize("mla", [False, True], ids=["mha", "mla"])
def test_default_layout_batches_its_prompts(ctx, mla):
    """The engine's default options (q2k_tiles = 1): decode multiplies planes, the batched path tile-record copies of the same
    weights (made when the first prompt arrives) - two float associations of the same integers, so the prompt's cache rows are the
    loop's within rounding, not bit for bit (that statement holds at q2k_tiles = 2; the oracle audit above holds at both levels).
    What this test pins: the call batches and the copies exist; the batched path on copies IS the batched path on stored tile
    records (cache rows of a level-2 model's dsk_hydrate: equal to an f16 place); block 0's K / V rows - upstream of every router -
    agree with the loop's to 2e-3 of their range; the last token's logits stay inside the K-quant free-running guard (5e-2 of their
    range: a random 3-block model amplifies a last-bit difference by ~100 per block, tests/test_teacher_forced_gpu.py says why)."""
    import dsk
    c = synth.preset("tiny_v3", "q2_k", mla)
    T = synth.synth_model(c, seed=61)
    tokens = [int(t) for t in np.random.default_rng(5).integers(0, c.vocab_size, 29)]
    A, B, C2 = dsk.Model(ctx, c, T), dsk.Model(ctx, c, T), dsk.Model(ctx, c, T, options={"q2k_tiles": 2})
    assert B.hydrate_why_not() == ""
    la, _ = _loop(A, tokens, 0)
    lb = B.hydrate(tokens, 0, dsk.MODE_OUTPUT_LOGITS)
    C2.hydrate(tokens, 0, dsk.MODE_OUTPUT_LOGITS)
    assert B.info("hydrate_batched_tokens") == len(tokens) and B.info("hydrate_looped_tokens") == 0
    assert B.info("hydrate_tile_copy_mb") >= 0 and A.info("hydrate_tile_copy_mb") == 0 and C2.info("hydrate_tile_copy_mb") == 0
    assert float(np.max(np.abs(la - lb))) < 5e-2 * float(np.max(np.abs(la)))

    def rel(a, b):
        a32, b32 = a.view(np.float16).astype(np.float32), b.view(np.float16).astype(np.float32)
        return float(np.max(np.abs(a32 - b32))) / max(1e-6, float(np.max(np.abs(a32))))
    ca, cb, cc = _caches(A, c, len(tokens)), _caches(B, c, len(tokens)), _caches(C2, c, len(tokens))
    assert rel(ca[0][0], cb[0][0]) < 2e-3 and rel(ca[0][1], cb[0][1]) < 2e-3
    for (kb, vb), (kc, vc) in zip(cb, cc):
        assert rel(kb, kc) < 2e-3 and rel(vb, vc) < 2e-3
    for M in (A, B, C2):
        M.close()


def test_hydrate_then_decode_continues_the_same_stream(ctx):
    """a prompt through dsk_hydrate, then greedy decoding with dsk_forward: the same tokens as the all-loop run"""
    import dsk
    c = synth.preset("tiny_v3", "q2_k", False)
    T = synth.synth_model(c, seed=53)
    prompt = [(13 * i + 1) % c.vocab_size for i in range(21)]
    outs = []
    for batched in (0, 1):
        M = dsk.Model(ctx, c, T, options={"q2k_tiles": 2, "hydrate_batched": batched})
        lo = M.hydrate(prompt, 0, dsk.MODE_OUTPUT_LOGITS)
        seq = []
        for i in range(12):
            t = int(np.argmax(lo))
            seq.append(t)
            lo = M.forward(t, len(prompt) + i)
        outs.append((seq, lo.copy(), M.info("hydrate_batched_tokens")))
        M.close()
    assert outs[0][2] == 0 and outs[1][2] == len(prompt)
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1])
