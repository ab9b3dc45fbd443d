"""Checkpoints written by the reference's OWN converter (tools/make_converter_fixture.py ran convert.py unmodified):
the header reader of the HIP loader (dsk_dseek_read_config, no GPU needed) must see the configuration convert.py
stringified (convert.py:123-170) exactly as Config::from_yalm would (src/model.cpp:21-127)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,quant,mla", [("q2k_mla", 3, 1), ("f8e5m2", 2, 0)])
def test_header_of_a_converter_made_checkpoint(name, quant, mla):
    import dsk
    d = os.path.join(GOLD, "converted_" + name)
    c, n_files, n_tensors, nbytes = dsk.read_dseek_config(d)
    assert n_files == 1 and n_tensors > 30 and nbytes > 1_000_000
    assert (c.dim, c.hidden_dim, c.n_layers, c.n_heads, c.vocab_size, c.max_seq_len) == (256, 512, 2, 4, 512, 64)
    assert (c.weight_quant, c.use_mla) == (quant, mla)
    assert (c.kv_lora_rank, c.q_lora_rank, c.qk_nope_head_dim, c.qk_rope_head_dim, c.v_head_dim) == (256, 256, 64, 32, 64)
    assert (c.n_routed_experts, c.n_active_routed, c.n_shared_experts, c.moe_intermediate_size) == (8, 2, 1, 256)
    assert (c.n_group, c.topk_group, c.first_k_dense_replace) == (4, 2, 1)
    assert c.norm_topk_prob == 1 and c.scoring_func == 1 and c.has_moegate_bias == 1
    assert c.topk_method == 1  # convert.py:113 maps noaux_tc to group_limited_greedy
    assert abs(c.routed_scaling_factor - 2.5) < 1e-6 and abs(c.norm_eps - 1e-6) < 1e-12
    assert tuple(c.block_size) == ((128, 128) if name == "f8e5m2" else (0, 0))
    g = np.load(os.path.join(GOLD, f"converted_{name}.npz"))
    assert g["seq_logits"].shape == (8, 512) and np.all(np.isfinite(g["seq_logits"]))
