"""Host-side logic that needs no GPU: the synthetic-checkpoint encoders (valid reference blocks),
the .dseek writer, the tensor-name -> role mapping, and the reference-format loader round trip."""
import json
import os
import struct
import tempfile

import numpy as np
import pytest

from tools import synth


@pytest.mark.parametrize("quant,enc,tol", [(3, synth.encode_q2k, 0.45), (4, synth.encode_q3k, 0.25)], ids=["q2_k", "q3_k"])
def test_encoders_produce_blocks_the_reference_layout_decodes(oracle, quant, enc, tol):
    rng = np.random.default_rng(0)
    w = rng.standard_normal((6, 1024)).astype(np.float32)
    wb = enc(w)
    assert wb.shape == (6, 4 * (84 if quant == 3 else 110)) and wb.dtype == np.uint8
    deq = np.stack([oracle.dequant_row(quant, wb[r], 1024) for r in range(6)])
    err = np.sqrt(np.mean((deq - w) ** 2)) / np.std(w)
    assert err < tol, err  # a wrong bit layout gives ~1.4


def test_f8_block_encoder(oracle):
    rng = np.random.default_rng(1)
    w = rng.standard_normal((200, 300)).astype(np.float32)
    b, s = synth.encode_f8_blocks(w, (128, 128))
    assert b.shape == (200, 300) and s.shape == (2, 3)
    deq = np.array([[oracle.lib.orc_f8e5m2_to_float(int(v)) for v in row] for row in b[:4]], np.float32)
    deq *= s[0, np.arange(300) // 128][None, :]
    assert np.max(np.abs(deq - w[:4])) < 0.3 * np.max(np.abs(w))
    assert not np.any((b & 0x7C) == 0x7C)  # no inf / nan encodings


def test_name_to_role_covers_every_tensor():
    for preset, quant, mla in (("tiny_v3", "q2_k", True), ("tiny_v3", "f8e5m2", False), ("tiny_v2lite", "q3_k", False)):
        c = synth.preset(preset, quant, mla)
        T = synth.synth_model(c, seed=3)
        seen = set()
        def bind(role, layer, q, shape, arr):
            assert (role, layer) not in seen
            seen.add((role, layer))
            assert arr.nbytes > 0 and shape[0] > 0
        synth.bind_all(T, bind)
        assert (synth.ROLE["EMBED"], -1) in seen and (synth.ROLE["WO"], c.n_layers - 1) in seen
        assert ((synth.ROLE["WC"], 0) in seen) == mla
        if quant == "f8e5m2":
            assert (synth.ROLE["W1"] + synth.ROLE["SCALE"], 1) in seen


def test_dseek_writer_layout():
    c = synth.preset("tiny_v2lite", "q2_k", False)
    T = synth.synth_model(c, seed=4)
    d = tempfile.mkdtemp()
    synth.write_dseek(d, c, T)
    raw = open(os.path.join(d, "shard_000.dseek"), "rb").read()
    (hl,) = struct.unpack("<Q", raw[:8])
    hdr = json.loads(raw[8:8 + hl])
    md = hdr["__metadata__"]
    assert md["quant"] == "q2_k" and md["arch"] == "DeepseekV2ForCausalLM" and all(isinstance(v, str) for v in md.values())
    e = hdr["model.layers.1.mlp.w1.weight"]  # K-quants are stored as U8 byte arrays (src/codec.cpp:170-207)
    assert e["dtype"] == "U8" and e["shape"] == [c.n_routed_experts, c.moe_intermediate_size, c.dim // 256 * 84]
    a, b = e["data_offsets"]
    assert raw[8 + hl + a:8 + hl + b] == T["model.layers.1.mlp.w1.weight"].data.tobytes()


def test_reference_loads_what_the_writer_wrote(ref, oracle):
    """The unmodified reference loader + forward accepts the synthetic checkpoint and agrees with the oracle."""
    c = synth.preset("tiny_v3", "fp16", False)
    T = synth.synth_model(c, seed=5)
    d = tempfile.mkdtemp()
    synth.write_dseek(d, c, T)
    S, M = ref.session(d, c), oracle.model(c, T)
    a, b = S.forward(7, 0), M.forward(7, 0)
    assert np.max(np.abs(a - b)) < 1e-4 * np.max(np.abs(a))
    assert S.active_bytes(0) > 0
    S.close()
    M.close()


def test_tile_planner_choices_for_the_deepseek_v3_shapes():
    """gemv_plan_tile (kernels_tile.hip) - the launches on Q2_K weights in the tiled layout (option "q2k_tiles", tile_device.h) -
    is host logic too: one 16-wave workgroup per CU for every launch that fills the chip, a round of 256 item partials (7168-wide
    rows: 7 four-block items per strip), LDS = the staged vector's 320-byte block records + the round's partials."""
    import dsk
    Q2, T = 3, 0x100
    rec = lambda n: n // 256 * 320
    # the routed experts' w1 / w3 (the default level tiles exactly these), the classifier, the dense and shared pairs
    for rows, nt, kind, act in ((2048, 9, 1, 0), (129280, 1, 0, 2), (18432, 1, 1, 2)):
        p = dsk.plan_gemv(Q2, rows, 7168, nt, kind | T, act)
        assert (p["waves"], p["grid"], p["groups"], p["rows_per_step"]) == (16, 256, 1, 256), (rows, p)
        assert p["lds_bytes"] == rec(7168) + 256 * 256, p
    # the two-launch experts' W2 with the fused combine: 8-wave workgroups, every task its own group with equal shares
    p = dsk.plan_gemv(Q2, 7168, 2048, 9, 3 | T, 1)
    assert (p["waves"], p["groups"], p["rows_per_step"]) == (8, 9, 128) and p["grid"] % 9 == 0 and p["grid"] >= 256, p
    assert p["lds_bytes"] == rec(2048) + 128 * 256, p
    # a launch with fewer strips than CUs gets one workgroup per strip (rows padded to 16): 1536 rows -> 96, 37 rows -> 3
    assert dsk.plan_gemv(Q2, 1536, 7168, 1, T)["grid"] == 96
    assert dsk.plan_gemv(Q2, 37, 512, 1, T, 1)["grid"] == 3
    # wo: 64 blocks per row -> 16 four-block items per strip, the vector's records dominate the LDS
    p = dsk.plan_gemv(Q2, 7168, 16384, 1, T, 0)
    assert (p["waves"], p["grid"]) == (16, 256) and p["lds_bytes"] == rec(16384) + 256 * 256, p
    with pytest.raises(dsk.DskError):
        dsk.plan_gemv(Q2, 64, 300, 1, T)      # not a multiple of 256 columns (quantizer.cpp:8)
    with pytest.raises(dsk.DskError):
        dsk.plan_gemv(4, 64, 512, 1, T)       # the tiled layout exists for Q2_K only


def test_launch_planner_choices_for_the_deepseek_v3_shapes():
    """gemv_plan (kernels_gemv.hip) is host logic: pin what it picks for the shapes of the headline model, so that a change
    of the rules shows up here and not as a silent slowdown (the measurements behind each choice: DESIGN.md 4.1 / 7)."""
    import dsk
    Q2, Q3, F8 = 3, 4, 2
    # first-stage projections: a small plain launch -> 64 lanes per row with a ragged second step, 16-wave workgroups
    p = dsk.plan_gemv(Q2, 1536, 7168)
    assert (p["lanes_per_row"], p["U"], p["waves"], p["grid"]) == (64, 2, 16, 256), p
    # big 7168-wide launches keep the exact fit of 16 lanes per row (7 column steps: the straight-line pipelined form)
    for rows, nt, kind in ((18432, 1, 1), (2048, 9, 1), (129280, 1, 0)):
        p = dsk.plan_gemv(Q2, rows, 7168, nt, kind)
        assert (p["lanes_per_row"], p["waves"], p["grid"], p["rows_per_step"]) == (16, 16, 256, 64), (rows, p)
    assert dsk.plan_gemv(Q2, 2048, 7168, 9, 1)["U"] == 4 and dsk.plan_gemv(Q3, 2048, 7168, 9, 1)["U"] == 2  # Q3_K: 4 steps spill
    # wo: 256 items = 4 steps of 64 lanes, one workgroup per CU; its input is the ready Q8_K vector (2.5 bytes / 2 values)
    p = dsk.plan_gemv(Q2, 7168, 16384, 1, 0, 0)
    assert (p["lanes_per_row"], p["U"], p["waves"], p["grid"]) == (64, 4, 16, 256) and p["lds_bytes"] == 16384 // 64 * 80, p
    # second-stage projections: 24 items -> 8 lanes per row, 128 rows per step
    p = dsk.plan_gemv(Q2, 24576, 1536)
    assert (p["lanes_per_row"], p["rows_per_step"], p["grid"]) == (8, 128, 256), p
    # the two-launch experts' W2 with the fused combine: one activation group per task, few tall row groups
    p = dsk.plan_gemv(Q2, 7168, 2048, 9, 3, 1)
    assert (p["lanes_per_row"], p["R"], p["waves"], p["groups"]) == (8, 2, 4, 9) and p["grid"] == 9 * 112, p
    # a K-quant row that no power-of-two lane count divides within the waste bound is refused, not mis-planned
    with pytest.raises(dsk.DskError):
        dsk.plan_gemv(Q2, 64, 300)
    with pytest.raises(dsk.DskError):
        dsk.plan_gemv(Q2, 64, 7168, 13)  # more tasks than a launch descriptor holds


def test_which_launches_request_their_weights_ahead_of_the_staging():
    """Round 6 (kernels_gemv.hip, option "gemv_ahead"): which planned launches run one of the "weights ahead of the staging"
    kernels at the engine's default options - host logic like the planner itself (gemv_ahead_kind behind dsk_plan_gemv_ahead), so a
    change of the planner's rules that silently sends the first-stage projections or wo back to the plain kernel shows up here."""
    import dsk
    Q2, Q3, F8 = 3, 4, 2
    ahead = dsk.plan_gemv_ahead
    # DeepSeek-V3: first-stage projections (wq_a || wkv_a: 2112 rows of 7168 on rmsnorm(x)), wo (7168 x 16384 on the ready Q8_K
    # vector), the MLA second stage (wq_rope_b || wc: 73 728 rows of 1536 on rmsnorm(q_a), launched with the cache-write rider)
    assert ahead(Q2, 2112, 7168, 1, 0, 2) == 1
    assert ahead(Q2, 1536, 7168, 1, 0, 2) == 1 and ahead(Q2, 576, 7168, 1, 0, 2) == 1
    assert ahead(Q2, 7168, 16384, 1, 0, 0) == 2
    assert ahead(Q2, 73728, 1536, 1, 0, 2, True) == 3
    # DeepSeek-V2-Lite: wq || wkv_a (3648 rows of 2048) and wo (2048 x 2048)
    assert ahead(Q2, 3648, 2048, 1, 0, 2) == 1
    assert ahead(Q2, 2048, 2048, 1, 0, 0) == 2
    # what does NOT qualify runs gemv_kernel: other quants, GLU pairs, the fused combine, rows of more column steps, a workgroup share
    # of more than one (f32 vector) or two (Q8_K vector) row groups, the MLA launch without its rider's plan
    assert ahead(Q3, 2112, 7168, 1, 0, 2) == 0 and ahead(F8, 2112, 7168, 1, 0, 2) == 0
    assert ahead(Q2, 18432, 7168, 1, 1, 2) == 0            # dense w1 / w3: a GLU pair
    assert ahead(Q2, 7168, 2048, 9, 3, 1) == 0             # the two-launch experts' W2 with the fused combine
    assert ahead(Q2, 129280, 7168, 1, 0, 2) == 0           # the classifier: 16 lanes per row, seven column steps
    assert ahead(Q2, 7168, 18432, 1, 0, 0) == 0            # dense w2: rows of 18432
    assert ahead(Q2, 24576, 1536, 1, 0, 2) == 0            # the MHA second stage is consumed per head, not launched like this
    assert ahead(Q2, 65536, 7168, 1, 0, 2) == 0            # a share of many row groups per workgroup
