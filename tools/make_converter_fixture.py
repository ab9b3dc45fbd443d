"""Checkpoints made by the REFERENCE's own converter (VERDICT r1 "missing" #4).

    python tools/make_converter_fixture.py            (needs /root/reference, torch, safetensors: this container)

Builds a tiny HuggingFace-layout DeepSeek-V3 directory (config.json, tokenizer.json, tokenizer_config.json,
model.safetensors; seeded Gaussian weights), runs the reference's `convert.py` on it UNMODIFIED --

    convert.py OUT IN --quant q2_k --mla       MLA weight absorption (wc = W_UK^T W_UQ, convert.py:384-438), per-expert
                                               K-quantisation + stacking (:344-362, 488-508), metadata strings (:123-170)
    convert.py OUT IN --quant f8e5m2           128 x 128 block scales (:262-275)

-- and records what the unmodified reference (oracle/_ref/libdskref.so) computes on the resulting `.dseek` files.  The
files land in tests/golden/converted_*/ with the recorded logits next to the directory (`converted_*.npz`);
tests/test_converted_gpu.py loads the same files with dsk_model_load_dseek and compares.

convert.py imports `quantizer_cpp`, a torch C++ extension around quantize_row_q2_K_ref / quantize_row_q3_K_ref
(quantizer.cpp).  Building it writes into the reference tree, which is read-only here; the stand-in module below calls
the very same two functions of the reference through oracle/_ref/libdskref.so (ref_quantize_row, oracle/ref_shim.cpp).
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

CONFIG = {
    "architectures": ["DeepseekV3ForCausalLM"], "hidden_size": 256, "intermediate_size": 512, "num_hidden_layers": 2,
    "num_attention_heads": 4, "vocab_size": 512, "bos_token_id": 0, "eos_token_id": 1, "rope_theta": 10000.0,
    "rms_norm_eps": 1e-6, "hidden_act": "silu", "first_k_dense_replace": 1, "kv_lora_rank": 256, "q_lora_rank": 256,
    "qk_nope_head_dim": 64, "qk_rope_head_dim": 32, "v_head_dim": 64, "n_shared_experts": 1, "n_routed_experts": 8,
    "num_experts_per_tok": 2, "moe_intermediate_size": 256, "routed_scaling_factor": 2.5, "n_group": 4,
    "norm_topk_prob": True, "scoring_func": "sigmoid", "topk_group": 2, "topk_method": "noaux_tc",
    "tie_word_embeddings": False,
    "rope_scaling": {"type": "yarn", "beta_fast": 32, "beta_slow": 1, "factor": 40, "mscale": 1.0, "mscale_all_dim": 1.0,
                     "original_max_position_embeddings": 4096},
}
TOKENS = [3, 17, 200, 511, 42, 99, 300, 7]

STANDIN = '''
import ctypes, numpy as np, torch
_L = ctypes.CDLL(%r)
_L.ref_quantize_row.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_int64]
def _q(t, quant, bsz):
    a = np.ascontiguousarray(t.detach().to(torch.float32).numpy())
    rows, n = a.shape
    out = np.zeros((rows, n // 256 * bsz), np.uint8)
    for r in range(rows):  # one quantize_row_q*_K_ref call per row, like quantizer.cpp:25-31
        _L.ref_quantize_row(quant, a[r].ctypes.data_as(ctypes.POINTER(ctypes.c_float)), out[r].ctypes.data, n)
    return torch.from_numpy(out)
def quantize_q2_k(t): return _q(t, 3, 84)
def quantize_q3_k(t): return _q(t, 4, 110)
'''


def hf_dir(d, seed):
    import torch
    from safetensors.torch import save_file
    from tools import synth
    c = CONFIG
    g = torch.Generator().manual_seed(seed)
    H, hd = c["num_attention_heads"], c["qk_nope_head_dim"] + c["qk_rope_head_dim"]
    dim, E, mi = c["hidden_size"], c["n_routed_experts"], c["moe_intermediate_size"]

    def W(*shape):
        return (torch.randn(*shape, generator=g) / shape[-1] ** 0.5).to(torch.bfloat16)  # HF checkpoints are bf16

    def N(n):
        return (1.0 + 0.1 * torch.randn(n, generator=g)).to(torch.bfloat16)

    w = {"model.embed_tokens.weight": torch.randn(c["vocab_size"], dim, generator=g).to(torch.bfloat16),
         "model.norm.weight": N(dim), "lm_head.weight": W(c["vocab_size"], dim)}
    for l in range(c["num_hidden_layers"]):
        p = f"model.layers.{l}."
        w[p + "input_layernorm.weight"] = N(dim)
        w[p + "post_attention_layernorm.weight"] = N(dim)
        w[p + "self_attn.kv_a_layernorm.weight"] = N(c["kv_lora_rank"])
        w[p + "self_attn.q_a_layernorm.weight"] = N(c["q_lora_rank"])
        w[p + "self_attn.q_a_proj.weight"] = W(c["q_lora_rank"], dim)
        w[p + "self_attn.q_b_proj.weight"] = W(H * hd, c["q_lora_rank"])
        w[p + "self_attn.kv_a_proj_with_mqa.weight"] = W(c["kv_lora_rank"] + c["qk_rope_head_dim"], dim)
        w[p + "self_attn.kv_b_proj.weight"] = W(H * (c["qk_nope_head_dim"] + c["v_head_dim"]), c["kv_lora_rank"])
        w[p + "self_attn.o_proj.weight"] = W(dim, H * c["v_head_dim"])
        if l < c["first_k_dense_replace"]:
            w[p + "mlp.gate_proj.weight"] = W(c["intermediate_size"], dim)
            w[p + "mlp.up_proj.weight"] = W(c["intermediate_size"], dim)
            w[p + "mlp.down_proj.weight"] = W(dim, c["intermediate_size"])
        else:
            w[p + "mlp.gate.weight"] = W(E, dim)
            w[p + "mlp.gate.e_score_correction_bias"] = (0.1 * torch.randn(E, generator=g)).to(torch.float32)
            for e in range(E):
                w[p + f"mlp.experts.{e}.gate_proj.weight"] = W(mi, dim)
                w[p + f"mlp.experts.{e}.up_proj.weight"] = W(mi, dim)
                w[p + f"mlp.experts.{e}.down_proj.weight"] = W(dim, mi)
            w[p + "mlp.shared_experts.gate_proj.weight"] = W(mi, dim)
            w[p + "mlp.shared_experts.up_proj.weight"] = W(mi, dim)
            w[p + "mlp.shared_experts.down_proj.weight"] = W(dim, mi)
    os.makedirs(d, exist_ok=True)
    save_file(w, os.path.join(d, "model.safetensors"))
    json.dump(c, open(os.path.join(d, "config.json"), "w"))
    json.dump({"model_max_length": 64}, open(os.path.join(d, "tokenizer_config.json"), "w"))
    vocab = synth.synthetic_vocab(c["vocab_size"])
    vocab = [t.replace(" ", "▁") for t in vocab]  # sentencepiece whitespace, undone by convert.py:201
    json.dump({"model": {"byte_fallback": True, "vocab": {t: i for i, t in enumerate(vocab)}}, "added_tokens": []},
              open(os.path.join(d, "tokenizer.json"), "w"))


def main():
    from oracle import orc
    if not os.path.exists(orc.REF_SO):
        orc.build(ref=True)
    work = tempfile.mkdtemp(prefix="dsk_convfix_")
    open(os.path.join(work, "quantizer_cpp.py"), "w").write(STANDIN % orc.REF_SO)
    hf = os.path.join(work, "hf")
    hf_dir(hf, seed=77)
    R = orc.Ref()
    R.set_threads(4)
    for name, args in (("q2k_mla", ["--quant", "q2_k", "--mla"]), ("f8e5m2", ["--quant", "f8e5m2"])):
        out = os.path.join(GOLD, "converted_" + name)
        shutil.rmtree(out, ignore_errors=True)
        env = dict(os.environ, PYTHONPATH=work + os.pathsep + os.environ.get("PYTHONPATH", ""))
        subprocess.check_call([sys.executable, os.path.join(REF, "convert.py"), out, hf] + args, env=env, stdout=subprocess.DEVNULL)
        # what the unmodified reference computes on these very files
        import dsk
        cfg = dsk.read_dseek_config(out)[0]
        S = R.session(out, cfg, context=0)
        seq = np.stack([S.forward(t, p) for p, t in enumerate(TOKENS)])
        ind = np.stack([S.forward(t, 0) for t in TOKENS])
        S.close()
        np.savez_compressed(out + ".npz",  # NEXT TO the directory: the loaders read every file inside it
                             tokens=np.array(TOKENS), seq_logits=seq, pos0_logits=ind)
        for f in os.listdir(out):
            os.chmod(os.path.join(out, f), 0o644)
        print(name, {f: os.path.getsize(os.path.join(out, f)) for f in sorted(os.listdir(out))})
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))
    main()
