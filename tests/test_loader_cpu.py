"""The `.dseek` loader's host half (include/dsk.h dsk_dseek_read_config, csrc/loader.cpp): directory scan, header JSON,
Config::from_yalm (src/model.cpp:21-127).  No GPU: only the headers are read.  The configuration parsed by the library
must equal, field by field, the one the tests build from the same Cfg object (dsk.make_config), and - where the
prebuilt reference is present - what the reference itself accepts from the same files (tests/test_model_gpu.py loads
the same directories through oracle/_ref)."""
import json
import os
import struct

import numpy as np
import pytest

from tools import synth


def _fields(d):
    return {n: (list(getattr(d, n)) if n == "block_size" else getattr(d, n)) for n, _ in d._fields_}


CASES = [("tiny_v3", "q2_k", False), ("tiny_v3", "q3_k", True), ("tiny_v3", "f8e5m2", False), ("tiny_v3", "fp16", True),
         ("tiny_v2lite", "q2_k", False), ("tiny_v2lite", "fp32", False)]


@pytest.mark.parametrize("preset,quant,mla", CASES, ids=[f"{a}-{b}-{'mla' if c else 'mha'}" for a, b, c in CASES])
def test_config_from_dseek_metadata_equals_make_config(tmp_path, preset, quant, mla):
    import dsk
    c = synth.preset(preset, quant, mla)
    T = synth.synth_model(c, seed=1)
    d = str(tmp_path / "ckpt")
    synth.write_dseek(d, c, T, shards=3)
    got, n_files, n_tensors, n_bytes = dsk.read_dseek_config(d)
    want = dsk.make_config(c)
    fg, fw = _fields(got), _fields(want)
    for k in fw:
        if isinstance(fw[k], float):
            assert fg[k] == pytest.approx(fw[k], rel=1e-7), k
        else:
            assert fg[k] == fw[k], k
    assert n_files == 3
    n_scale = sum(1 for t in T.values() if t.scale is not None)
    assert n_tensors == len(T) + n_scale
    assert n_bytes == sum(t.data.nbytes + (t.scale.nbytes if t.scale is not None else 0) for t in T.values())
    # the reference's -c option caps max_seq_len (src/model.cpp:73-76)
    assert dsk.read_dseek_config(d, context=7)[0].max_seq_len == min(7, c.max_seq_len)


def _write_raw(d, header: dict, payload: bytes = b""):
    os.makedirs(d, exist_ok=True)
    hj = json.dumps(header).encode()
    with open(os.path.join(d, "a.dseek"), "wb") as f:
        f.write(struct.pack("<Q", len(hj)) + hj + payload)


def test_loader_rejects_what_the_reference_rejects(tmp_path):
    import dsk
    c = synth.preset("tiny_v3", "q2_k", False)
    md = c.metadata()
    with pytest.raises(dsk.DskError, match="cannot open directory"):
        dsk.read_dseek_config(str(tmp_path / "nope"))
    os.makedirs(tmp_path / "empty")
    with pytest.raises(dsk.DskError, match="no files"):
        dsk.read_dseek_config(str(tmp_path / "empty"))
    # header length beyond the file (src/codec.cpp:303-306)
    d = str(tmp_path / "trunc")
    os.makedirs(d)
    open(os.path.join(d, "a.dseek"), "wb").write(struct.pack("<Q", 1000) + b"{}")
    with pytest.raises(dsk.DskError, match="no valid header"):
        dsk.read_dseek_config(d)
    # not JSON
    d = str(tmp_path / "garbage")
    os.makedirs(d)
    open(os.path.join(d, "a.dseek"), "wb").write(struct.pack("<Q", 5) + b"[1,2]")
    with pytest.raises(dsk.DskError, match="not a JSON object"):
        dsk.read_dseek_config(d)
    # a required key is missing (std::map::at throws in the reference)
    m2 = dict(md)
    del m2["rope_theta"]
    _write_raw(str(tmp_path / "nokey"), {"__metadata__": m2})
    with pytest.raises(dsk.DskError, match="rope_theta"):
        dsk.read_dseek_config(str(tmp_path / "nokey"))
    # unsupported quant / topk_method (the reference asserts)
    _write_raw(str(tmp_path / "q"), {"__metadata__": dict(md, quant="q4_k")})
    with pytest.raises(dsk.DskError, match="unsupported quant"):
        dsk.read_dseek_config(str(tmp_path / "q"))
    _write_raw(str(tmp_path / "tk"), {"__metadata__": dict(md, topk_method="noaux_tc")})
    with pytest.raises(dsk.DskError, match="noaux_tc"):
        dsk.read_dseek_config(str(tmp_path / "tk"))
    # tensor entries: size / shape disagreement, offsets past the data, bad dtype (src/codec.cpp:124-165)
    ok = {"dtype": "F32", "shape": [2, 2], "data_offsets": [0, 16]}
    _write_raw(str(tmp_path / "t_ok"), {"__metadata__": md, "x": ok}, b"\0" * 16)
    assert dsk.read_dseek_config(str(tmp_path / "t_ok"))[2] == 1
    for name, bad, msg in (("t_size", dict(ok, shape=[3, 2]), "shape and size"),
                           ("t_off", dict(ok, data_offsets=[0, 64]), "bad offsets"),
                           ("t_dt", dict(ok, dtype="F64"), "bad dtype")):
        _write_raw(str(tmp_path / name), {"__metadata__": md, "x": bad}, b"\0" * 16)
        with pytest.raises(dsk.DskError, match=msg):
            dsk.read_dseek_config(str(tmp_path / name))


def test_defaults_of_optional_metadata_keys(tmp_path):
    """Keys the reference reads with a default (src/model.cpp:28-35, 57-63, 78-79, 81, 99-100)."""
    import dsk
    md = synth.preset("tiny_v3", "fp16", False).metadata()
    for k in ("n_shared_experts", "n_routed_experts", "n_active_routed", "moe_intermediate_size", "routed_scaling_factor", "n_group",
              "norm_topk_prob", "scoring_func", "topk_group", "topk_method", "use_mla", "kv_lora_rank", "q_lora_rank", "norm_eps",
              "act_type", "first_k_dense_replace"):
        md.pop(k, None)
    md["arch"] = "DeepseekV2ForCausalLM"
    _write_raw(str(tmp_path / "d"), {"__metadata__": md})
    c = dsk.read_dseek_config(str(tmp_path / "d"))[0]
    assert (c.n_shared_experts, c.n_routed_experts, c.n_active_routed, c.moe_intermediate_size) == (0, 0, 0, 0)
    assert c.routed_scaling_factor == 1.0 and c.n_group == 1 and c.norm_topk_prob == 0 and c.topk_group == 0
    assert c.scoring_func == 0 and c.topk_method == 0 and c.use_mla == 0 and c.has_moegate_bias == 0
    assert c.kv_lora_rank == 0 and c.q_lora_rank == 0 and c.first_k_dense_replace == 0
    assert c.norm_eps == pytest.approx(1e-5) and c.act == 0  # gelu


def test_header_json_dialect(tmp_path):
    """The header is whatever a JSON writer emits: whitespace, escapes, unicode escapes, numbers as floats, nested values
    the loader does not use; a later shard's tensor of the same name replaces an earlier one (the reference assigns into
    a std::map, src/codec.cpp:318-327)."""
    import dsk
    md = synth.preset("tiny_v3", "fp16", False).metadata()
    md["note"] = "tab\there é \"quoted\" back\\slash"
    d = str(tmp_path / "d")
    os.makedirs(d)
    hdr = {"__metadata__": md, "a/b\"c": {"dtype": "F16", "shape": [2, 3.0], "data_offsets": [0, 12], "extra": {"nested": [1, {"x": None}, True]}},
           "second": {"dtype": "U8", "shape": [4], "data_offsets": [12, 16]}}
    hj = json.dumps(hdr, indent=3, ensure_ascii=True).encode()  # \\u00e9 escape, newlines and indentation
    open(os.path.join(d, "000.dseek"), "wb").write(struct.pack("<Q", len(hj)) + hj + b"\0" * 16)
    hj2 = json.dumps({"second": {"dtype": "F32", "shape": [2], "data_offsets": [0, 8]}}, separators=(",", ":")).encode()
    open(os.path.join(d, "001.dseek"), "wb").write(struct.pack("<Q", len(hj2)) + hj2 + b"\0" * 8)
    c, n_files, n_tensors, n_bytes = dsk.read_dseek_config(d)
    assert (n_files, n_tensors) == (2, 2) and c.dim == synth.preset("tiny_v3", "fp16", False).dim
    assert n_bytes == 12 + 4 + 8  # every tensor entry read counts, also the replaced one
    # a shape entry that is not an integer, a negative offset
    for bad in ({"dtype": "F32", "shape": [1.5], "data_offsets": [0, 6]}, {"dtype": "U8", "shape": [4], "data_offsets": [-4, 0]}):
        dd = str(tmp_path / ("bad%d" % id(bad)))
        _write_raw(dd, {"__metadata__": md, "t": bad}, b"\0" * 16)
        with pytest.raises(dsk.DskError):
            dsk.read_dseek_config(dd)
