"""Teacher-forced, flip-audited parity of the HIP engine, block by block, through the C ABI (tests/teacher.py).

Every block of a model is run on the ORACLE's residual stream x_{l-1} (dsk_model_run_block), every Q8_K staging point
is tapped, and the block is proven equal to the reference's arithmetic stage by stage: int8 codes identical except
proven rounding ties (counted), integer GEMVs on the device's own codes within 2e-5, expert indices identical, the
block output within 2e-6 of the k-ordered combine of the device's own slot outputs.  The free-running comparison of
x_l against the oracle is reported next to it and held to the north star's 1e-3 whenever the block had no flip.

Full width: DeepSeek-V3 shapes (dim 7168, 128 heads, vocab 129280), 256 routed experts, n_group 8 / topk_group 4,
top-8, 1 dense + 1 MoE block, MHA and MLA.
"""
import numpy as np
import pytest

from tests import teacher
from tests.util import MODEL_CASES, case_id, is_kquant, rel_inf
from tools import synth

pytestmark = pytest.mark.gpu


def _free_running_check(A, x_hip, x_orc, what, pos):
    """x_l of the device against the ORACLE'S OWN x_l on the same x_{l-1}: reported, and capped at 5e-2 as a guard against
    gross errors.  It cannot be held to 1e-3 block by block: the oracle's later staging points see ITS float inputs,
    which differ from the device's in the last bits, so a rounding tie can fall differently there even when every device
    code equals the oracle's quantisation of the device's own vector (no flip in the audit's sense); at pos > 0 the
    cache rows earlier calls wrote carry such differences on.  The 1e-3 statement is the audit's: the block output
    equals the reference arithmetic on the device's codes to 2e-6 (teacher.py, stage x_out)."""
    e = rel_inf(x_hip, x_orc)
    assert e < 5e-2, (what, e, A.summary())
    return e


KQ_CASES = [c for c in MODEL_CASES if is_kquant(c[1])]
# (case, option q2k_tiles): every K-quant golden model at the default layout, and the Q2_K ones with EVERY converted role as tile
# records (level 2: the tiled second stage of the MLA launch, wv_b on the matrix pipe inside mla_head_kernel, the tiled first
# stage / per-head projections / wo / dense FFN / classifier) - the layout dsk_hydrate's batched path and the seam run on
KQ_TILE_CASES = [(c, None) for c in KQ_CASES] + [(c, 2) for c in KQ_CASES if c[1] == "q2_k" and c[0] == "tiny_v3"]


@pytest.mark.parametrize("case,tiles", KQ_TILE_CASES, ids=[case_id(c) + ("" if t is None else f"-tiles{t}") for c, t in KQ_TILE_CASES])
def test_every_block_teacher_forced_on_the_oracle_stream(ctx, oracle, case, tiles):
    import dsk
    preset, quant, mla, seed = case
    c = synth.preset(preset, quant, mla)
    T = synth.synth_model(c, seed=seed)
    M, O = dsk.Model(ctx, c, T, options=None if tiles is None else {"q2k_tiles": tiles}), oracle.model(c, T)
    if tiles:
        assert M.info("tiled_tensors") > 0
    aud = teacher.BlockAuditor(oracle, c, T)
    emb = T["model.embed.weight"]
    flips, worst, free = 0, 0.0, []
    for pos, tok in enumerate([5, 77, 300, 901, 12]):
        lo = O.forward(tok, pos)
        x = oracle.embed_row(emb.quant, emb.data, c.dim, tok)
        for l in range(c.n_layers):
            A, x_hip = aud.run(M, l, x, pos)
            flips += A.total_flips()
            worst = max(worst, max(A.errs.values()))
            x_orc = O.trace_x(l)
            free.append(_free_running_check(A, x_hip, x_orc, (case_id(case), pos, l), pos))
            x = x_orc  # teacher forcing: the next block sees the oracle's stream
        A, logits = teacher.audit_head(oracle, c, T, M, x)
        flips += A.total_flips()
        assert rel_inf(logits, lo) < 5e-2
    print(f"\n[{case_id(case)}] {5 * c.n_layers} blocks: worst stage error {worst:.2e}, {flips} near-tie flips, "
          f"free-running x_l error median {np.median(free):.2e} max {max(free):.2e}, within 1e-3 in {sum(e < 1e-3 for e in free)}/{len(free)} blocks")
    assert np.median(free) < 1e-3  # most blocks see no tie at all
    # the taps are off again after run_block: a normal forward still matches a flip-free oracle token
    M.close()
    O.close()


def _v3_full_width(mla, seed, quant="q2_k"):
    c = synth.preset("v3", quant, mla, n_layers=2, first_k_dense_replace=1, max_seq_len=64)
    assert c.n_routed_experts == 256 and c.n_group == 8 and c.topk_group == 4 and c.n_active_routed == 8
    T = synth.random_block_model(c, seed=seed, tile_blocks=(1 << 21) + 12345)
    rng = np.random.default_rng(seed + 100)
    for name, t in T.items():  # norms and gate bias like a real checkpoint, not all-ones
        if name.endswith("norm.weight"):
            t.data = (1.0 + 0.1 * rng.standard_normal(t.data.size)).astype(np.float32)
        if name.endswith("moegate.bias"):
            t.data = (0.1 * rng.standard_normal(t.data.size)).astype(np.float32)
    return c, T


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("mla,quant,tiles", [(False, "q2_k", None), (True, "q2_k", None), (False, "q3_k", None), (False, "q2_k", 2), (False, "q2_k", 0), (True, "q2_k", 2)],
                         ids=["mha", "mla", "mha-q3_k", "mha-all-tiles", "mha-no-tiles", "mla-all-tiles"])
def test_v3_full_width_256_experts_teacher_forced(ctx, oracle, mla, quant, tiles):
    """BASELINE.json configs[3] at full width: 256 routed experts, 8 groups / 4 kept, top-8; 1 dense + 1 MoE block.  Q3_K at
    the same width runs the other instantiation of every K-quant kernel (moe_ffn_kernel<Q3_K, 2, 2>, the generic row loops).
    `tiles`: option q2k_tiles - default (the experts' matrices as tile records, matrix-pipe row products), 2 (every converted role:
    first-stage projections, the per-head attention launch's projections, wo, dense FFN, classifier), 0 (planes + dot4 everywhere)."""
    import dsk
    c, T = _v3_full_width(mla, seed=31, quant=quant)
    M, O = dsk.Model(ctx, c, T, options=None if tiles is None else {"q2k_tiles": tiles}), oracle.model(c, T)
    if quant == "q2_k":
        assert (M.info("tiled_tensors") > 0) == (tiles != 0)
    aud = teacher.BlockAuditor(oracle, c, T)
    emb = T["model.embed.weight"]
    flips, worst, free, routes = 0, 0.0, [], []
    x = None
    for pos, tok in enumerate([3, 99999, 123456]):
        O.forward(tok, pos, mode=0)  # hydrate: the oracle's stream and caches up to this position
        x = oracle.embed_row(emb.quant, emb.data, c.dim, tok)
        for l in range(c.n_layers):
            A, x_hip = aud.run(M, l, x, pos)
            flips += A.total_flips()
            worst = max(worst, max(A.errs.values()))
            x_orc = O.trace_x(l)
            free.append(_free_running_check(A, x_hip, x_orc, ("v3", "mla" if mla else "mha", pos, l), pos))
            if l >= c.first_k_dense_replace:
                e_dev = M.stage("route_e", c.n_active_routed, np.int32)
                routes.append(bool(np.array_equal(e_dev, O.routing()[0][l])))
                assert len(set(e_dev.tolist())) == c.n_active_routed and e_dev.min() >= 0 and e_dev.max() < 256
            x = x_orc
        print(f"\n[v3 {'mla' if mla else 'mha'} pos {pos}] {A.summary()}")
    # classifier: final norm + Q8_K + 129 280 x 7168 GEMV on a sample of rows (rows are independent)
    rows = np.unique(np.concatenate([[0, 1, c.vocab_size - 1], np.random.default_rng(5).integers(0, c.vocab_size, 253)]))
    A, _ = teacher.audit_head(oracle, c, T, M, x, rows)
    print(f"[v3 {'mla' if mla else 'mha'}] worst stage error {worst:.2e}; {flips} proven near-tie flips; free-running x_l error "
          f"{[f'{e:.1e}' for e in free]}; routing equal to the free-running oracle in {sum(routes)}/{len(routes)} MoE blocks; head {A.summary()}")
    assert worst < teacher.FLOAT_TOL
    # the blocks were fed the ORACLE's stream, so the free-running oracle routes on (almost) the same input: at most one
    # of the MoE blocks may differ (a proven tie upstream moves a router logit by ~1e-4 of its scale)
    assert sum(routes) >= len(routes) - 1, routes
    M.close()
    O.close()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("level", [2, 1], ids=["tiles-everywhere", "default-layout"])
@pytest.mark.parametrize("mla", [False, True], ids=["mha", "mla"])
def test_v3_full_width_batched_prompt_audited_on_the_oracle(ctx, oracle, mla, level):
    """The batched prompt path (dsk_hydrate, src/main.cpp:312-319 as GEMMs) against the ORACLE at full DeepSeek-V3 width
    (256 experts, 128 heads, 1 dense + 1 MoE block), not against the engine's own loop: a 37-token prompt in one chunk, then for
    the first, a middle and the last token every stage of every block goes through the same audit as the per-token block
    (teacher.HydrateDevice: codes = the oracle's except proven ties, integer GEMVs and float stages on the device's codes within
    2e-5 / 1e-4, the position's K/V (latent) cache row to the last f16 place, attention over the rows the SAME chunk wrote,
    expert indices identical on the device's router logits, the k-ordered combine within 2e-6).
    `default-layout` (round 6): the engine's DEFAULT options - decode keeps every matrix but the experts' as planes, the batched
    path multiplies tile-record copies made when the first prompt arrives (option "hydrate_tile_copies") - held to the same audit:
    the model bench.py times is a model that batches its prompts."""
    import dsk
    c, T = _v3_full_width(mla, seed=33)
    emb = T["model.embed.weight"]
    tokens = [int(t) for t in np.random.default_rng(3).integers(0, c.vocab_size, 37)]
    aud = teacher.BlockAuditor(oracle, c, T)
    worst, flips = 0.0, 0
    for l in range(c.n_layers):
        M = dsk.Model(ctx, c, T, options={"q2k_tiles": level, "hydrate_tap_layer": l})
        assert M.hydrate_why_not() == ""
        M.set_trace(True)
        M.hydrate(tokens, 0, dsk.MODE_HYDRATE_KV_CACHE)
        assert M.info("hydrate_batched_tokens") == len(tokens) and M.info("hydrate_looped_tokens") == 0
        assert (M.info("hydrate_tile_copy_mb") > 0) == (level == 1)
        for i in (0, 18, 36):
            x_in = oracle.embed_row(emb.quant, emb.data, c.dim, tokens[i]) if l == 0 else M.hydrate_trace_x(l - 1, i)
            A, _ = aud.run(teacher.HydrateDevice(M, c, l, i), l, x_in, i)
            flips += A.total_flips()
            worst = max(worst, max(A.errs.values()))
            print(f"\n[v3 {'mla' if mla else 'mha'} batched prompt, q2k_tiles={level}, block {l}, token {i}] {A.summary()}")
        M.close()
    assert worst < teacher.FLOAT_TOL


@pytest.mark.timeout(1500)
def test_v3_full_width_batched_prompt_in_the_mla_matrix_core_regime_audited_on_the_oracle(ctx, oracle):
    """Round 6: dsk_hydrate batches MLA tokens whose context has reached mla_flash_min_kv (320) - decode's mla_flash_kernel over the
    chunk's tokens + the per-head merge (hydrate.cpp hyd_layer).  Oracle-side, at full DeepSeek-V3 width: caches pre-filled with 400
    random f16 rows per block, a 24-token chunk at positions 400-423 (every token in the matrix-core regime; two flash launches), and
    for the first, a middle and the last token of the chunk every stage of every block through the block audit - this position's
    latent / rope row to the last f16 place, latent_out against the oracle's attn_mla (src/infer.cpp:766-804) over the 401-424 rows
    the device holds, the Q8_K of it, wv_b, wo, router, experts."""
    import dsk
    c, T = _v3_full_width(True, seed=41)
    c.max_seq_len = 512
    emb = T["model.embed.weight"]
    tokens = [int(t) for t in np.random.default_rng(4).integers(0, c.vocab_size, 24)]
    aud = teacher.BlockAuditor(oracle, c, T)
    lora, rope = c.kv_lora_rank, c.qk_rope_head_dim
    worst, flips = 0.0, 0
    for l in range(c.n_layers):
        M = dsk.Model(ctx, c, T, options={"hydrate_tap_layer": l})
        rng = np.random.default_rng(78)
        for ll in range(c.n_layers):
            M.set_cache_rows(ll, "nope_cache", 0, _f16_bits(rng.standard_normal((400, lora))))
            M.set_cache_rows(ll, "rope_cache", 0, _f16_bits(rng.standard_normal((400, rope))))
        assert M.hydrate_why_not() == ""
        M.set_trace(True)
        M.hydrate(tokens, 400, dsk.MODE_HYDRATE_KV_CACHE)
        assert M.info("hydrate_batched_tokens") == len(tokens) and M.info("hydrate_looped_tokens") == 0
        for i in (0, 11, 23):
            x_in = oracle.embed_row(emb.quant, emb.data, c.dim, tokens[i]) if l == 0 else M.hydrate_trace_x(l - 1, i)
            A, _ = aud.run(teacher.HydrateDevice(M, c, l, i), l, x_in, 400 + i)
            flips += A.total_flips()
            worst = max(worst, max(A.errs.values()))
            print(f"\n[v3 mla batched prompt at pos {400 + i}, block {l}] latent_out {A.errs['latent_out']:.2e}; {A.summary()}")
        M.close()
    assert worst < teacher.FLOAT_TOL


def _f16_bits(a):
    return np.asarray(a, np.float32).astype(np.float16).view(np.uint16)


@pytest.mark.timeout(2400)
@pytest.mark.parametrize("mla", [False, True], ids=["mha", "mla"])
def test_v3_full_width_long_context_regimes_teacher_forced(ctx, oracle, mla):
    """the attention regimes that only exist at long contexts, audited at full DeepSeek-V3 width
    (128 heads, Q2_K, 1 dense + 1 MoE block with 256 experts) - the kernels bench.py's kv_sweep times:
      MLA  kv_len >= 320 : scores / values of all heads on the matrix cores (mla_flash_kernel) + per-head merge
      MHA  kv_len >= 1024: two workgroups per head over halves of the context, merged by the last (head_attn_kernel)
      both at the full ring (pos >= 4096: kv_len 4096, the cache row lands at kv_pos, two sink keys re-rotated).
    The caches are pre-filled with random f16 rows (dsk_model_set_cache_rows); the audit then proves, on the device's own
    cache rows: this position's row to the last f16 place, attention (att_out / latent_out) within FLOAT_TOL of the
    oracle's attn / attn_mla (src/infer.cpp:728-804), the Q8_K of the attention output bit-exact, and every later stage
    of the block as at short contexts."""
    import dsk
    c, T = _v3_full_width(mla, seed=37)
    c.max_seq_len = 4200
    W = c.rs_original_max_position_embeddings
    assert W == 4096
    M = dsk.Model(ctx, c, T)
    aud = teacher.BlockAuditor(oracle, c, T)
    rng = np.random.default_rng(77)
    H, hd, vd, lora, rope = c.n_heads, c.head_dim, c.v_head_dim, c.kv_lora_rank, c.qk_rope_head_dim
    for l in range(c.n_layers):
        for r0 in range(0, 4200, 700):  # pieces: a whole MHA cache is 200 MB of f16
            n = min(700, 4200 - r0)
            if mla:
                M.set_cache_rows(l, "nope_cache", r0, _f16_bits(rng.standard_normal((n, lora))))
                M.set_cache_rows(l, "rope_cache", r0, _f16_bits(rng.standard_normal((n, rope))))
            else:
                M.set_cache_rows(l, "k_cache", r0, _f16_bits(0.5 * rng.standard_normal((n, H * hd), dtype=np.float32)))
                M.set_cache_rows(l, "v_cache", r0, _f16_bits(rng.standard_normal((n, H * vd), dtype=np.float32)))
    emb = T["model.embed.weight"]
    worst, flips = 0.0, 0
    for pos, tok in ((329, 11), (1029, 70000), (4000, 5), (4100, 129279)):
        x = oracle.embed_row(emb.quant, emb.data, c.dim, tok)
        for l in range(c.n_layers):
            A, x = aud.run(M, l, x, pos)
            flips += A.total_flips()
            worst = max(worst, max(A.errs.values()))
            key = "latent_out" if mla else "att_out"
            print(f"\n[v3 {'mla' if mla else 'mha'} pos {pos} layer {l}] {key} {A.errs[key]:.2e}; {A.summary()}")
    assert worst < teacher.FLOAT_TOL
    M.close()


def test_router_logits_op_vs_sequential_reference_sum(ctx, oracle):
    """the F32 router GEMV (src/infer.cpp:847, :121-157: a strictly sequential f32 sum in the reference
    binary) as the router kernel computes it (column slices, fixed tree): 256 x 7168 and a ragged small shape, with and
    without the fused rmsnorm."""
    rng = np.random.default_rng(9)
    for E, dim in ((256, 7168), (64, 2048), (16, 512), (7, 1000)):
        w = (rng.standard_normal((E, dim)) / np.sqrt(dim)).astype(np.float32)
        x = (rng.standard_normal(dim) * 3).astype(np.float32)
        nw = (1.0 + 0.1 * rng.standard_normal(dim)).astype(np.float32)
        got = ctx.router_logits(w, x, None)
        ref = oracle.gemv(0, w, E, dim, x)
        assert np.max(np.abs(got - ref)) < 2e-5 * max(1.0, float(np.max(np.abs(ref)))), (E, dim, float(np.max(np.abs(got - ref))))
        got = ctx.router_logits(w, x, nw, 1e-6)
        ref = oracle.gemv(0, w, E, dim, oracle.rmsnorm(x, nw, 1e-6))
        assert np.max(np.abs(got - ref)) < 2e-5 * max(1.0, float(np.max(np.abs(ref)))), (E, dim, float(np.max(np.abs(got - ref))))
        assert np.array_equal(got, ctx.router_logits(w, x, nw, 1e-6))  # fixed tree: bit-reproducible


def test_run_block_and_get_stage_reject_bad_arguments(ctx):
    """the harness entry points return error codes like the rest of the boundary (never abort)"""
    import dsk
    c = synth.preset("tiny_v3", "q2_k", False)
    M = dsk.Model(ctx, c, None, synth_seed=1)
    x = np.zeros(c.dim, np.float32)
    with pytest.raises(dsk.DskError):
        M.stage("x_mid", c.dim)          # nothing tapped yet
    with pytest.raises(dsk.DskError):
        M.run_block(c.n_layers, x, 0)    # no such layer
    with pytest.raises(dsk.DskError):
        M.run_block(0, x, -1)
    M.run_block(1, x, 0)
    with pytest.raises(dsk.DskError):
        M.stage("no_such_stage", 4)
    with pytest.raises(dsk.DskError):
        M.stage("x_mid", c.dim + 1)      # more than the stage holds
    assert M.stage("x_mid", c.dim).shape == (c.dim,)
    # a normal decode step still works (graph replay) and the taps are off: bit-identical to a fresh model
    a = M.forward(7, 0)
    M2 = dsk.Model(ctx, c, None, synth_seed=1)
    assert np.array_equal(a, M2.forward(7, 0))
    M.close()
    M2.close()
