"""Synthetic DeepSeek checkpoints for parity tests and benches.

No real weights exist on the build or GPU machines, so every input is synthetic:
Gaussian float weights are encoded into the reference's on-disk formats with the small numpy
encoders below (valid Q2_K / Q3_K / F8E5M2-block / F16 bytes -- they do NOT have to match the
reference's offline quantizer, only its *decode-side* layout: src/quant.h:41-52,70-76 and
src/quant.cpp:207-211,361-378; convert.py:262-275 for the F8 block scales).

Tensors are named exactly like the reference's `.dseek` shards (src/model.cpp:766-871) so that
`write_dseek` produces a directory the unmodified reference loads, and `bind_all` feeds the very
same bytes to the oracle and to the HIP engine.
"""
from __future__ import annotations

import json
import os
import struct
from dataclasses import dataclass, field
from typing import Dict, Tuple

import numpy as np

QK_K = 256
Q2K_BYTES, Q3K_BYTES = 84, 110
QUANT_IDS = {"fp32": 0, "fp16": 1, "f8e5m2": 2, "q2_k": 3, "q3_k": 4}
QUANT_NAMES = {v: k for k, v in QUANT_IDS.items()}

# dsk_role (include/dsk.h)
ROLE = dict(
    EMBED=0, FINAL_NORM=1, OUTPUT=2, ATTN_NORM=10, Q_A_NORM=11, KV_A_NORM=12, FFN_NORM=13, WQ=14,
    WQ_A=15, WQ_B=16, WKV_A=17, WKV_B=18, WO=19, WC=20, WQ_ROPE_B=21, WV_B=22, W1=23, W2=24, W3=25,
    SHARED_W1=26, SHARED_W2=27, SHARED_W3=28, MOEGATE=29, MOEGATE_BIAS=30, SCALE=64,
)
_LAYER_NAMES = {
    "attn.norm": "ATTN_NORM", "attn.q_a_norm": "Q_A_NORM", "attn.kv_a_norm": "KV_A_NORM",
    "mlp.norm": "FFN_NORM", "attn.wq": "WQ", "attn.wq_a": "WQ_A", "attn.wq_b": "WQ_B",
    "attn.wkv_a": "WKV_A", "attn.wkv_b": "WKV_B", "attn.wo": "WO", "attn.wc": "WC",
    "attn.wq_rope_b": "WQ_ROPE_B", "attn.wv_b": "WV_B", "mlp.w1": "W1", "mlp.w2": "W2", "mlp.w3": "W3",
    "shared_mlp.w1": "SHARED_W1", "shared_mlp.w2": "SHARED_W2", "shared_mlp.w3": "SHARED_W3",
    "moegate": "MOEGATE",
}


@dataclass
class Cfg:
    """Mirror of dsk_config (include/dsk.h) / reference Config (src/model.h:47-96)."""
    arch: str = "DeepseekV3ForCausalLM"
    dim: int = 512
    hidden_dim: int = 512
    n_layers: int = 3
    n_heads: int = 4
    vocab_size: int = 1024
    max_seq_len: int = 64
    rope_theta: float = 10000.0
    norm_eps: float = 1e-6
    act: str = "silu"
    first_k_dense_replace: int = 1
    n_shared_experts: int = 1
    n_routed_experts: int = 16
    n_active_routed: int = 4
    moe_intermediate_size: int = 256
    routed_scaling_factor: float = 2.5
    n_group: int = 4
    norm_topk_prob: bool = True
    scoring_func: str = "sigmoid"
    topk_group: int = 2
    topk_method: str = "group_limited_greedy"
    use_mla: bool = False
    kv_lora_rank: int = 256
    q_lora_rank: int = 256
    qk_nope_head_dim: int = 64
    qk_rope_head_dim: int = 32
    v_head_dim: int = 64
    quant: str = "q2_k"
    block_size: Tuple[int, int] = (0, 0)
    rs_original_max_position_embeddings: int = 4096
    tie_embeddings: bool = False

    @property
    def head_dim(self):
        return self.qk_nope_head_dim + self.qk_rope_head_dim

    @property
    def has_moegate_bias(self):
        return self.arch == "DeepseekV3ForCausalLM"

    def metadata(self) -> Dict[str, str]:
        """String->string metadata as convert.py:123-170 writes it."""
        md = {
            "arch": self.arch, "use_mla": str(int(self.use_mla)), "quant": self.quant,
            "dim": self.dim, "hidden_dim": self.hidden_dim, "n_layers": self.n_layers,
            "n_heads": self.n_heads, "vocab_size": self.vocab_size, "max_seq_len": self.max_seq_len,
            "bos_token_id": 0, "eos_token_id": 1, "rope_theta": self.rope_theta,
            "norm_eps": repr(self.norm_eps), "norm_type": "rmsnorm", "act_type": self.act,
            "first_k_dense_replace": self.first_k_dense_replace, "kv_lora_rank": self.kv_lora_rank,
            "q_lora_rank": self.q_lora_rank, "qk_nope_head_dim": self.qk_nope_head_dim,
            "qk_rope_head_dim": self.qk_rope_head_dim, "v_head_dim": self.v_head_dim,
            "n_shared_experts": self.n_shared_experts, "n_routed_experts": self.n_routed_experts,
            "n_active_routed": self.n_active_routed, "moe_intermediate_size": self.moe_intermediate_size,
            "routed_scaling_factor": self.routed_scaling_factor, "n_group": self.n_group,
            "norm_topk_prob": str(bool(self.norm_topk_prob)), "scoring_func": self.scoring_func,
            "topk_group": self.topk_group, "topk_method": self.topk_method,
            "rope_scaling_beta_fast": 32, "rope_scaling_beta_slow": 1, "rope_scaling_factor": 40,
            "rope_scaling_mscale": 1.0, "rope_scaling_mscale_all_dim": 1.0,
            "rope_scaling_original_max_position_embeddings": self.rs_original_max_position_embeddings,
        }
        if self.quant == "f8e5m2":
            md["quantization_block_size_0"] = self.block_size[0]
            md["quantization_block_size_1"] = self.block_size[1]
        return {k: str(v) for k, v in md.items()}


# ---------------------------------------------------------------- presets
def preset(name: str, quant: str = "q2_k", use_mla: bool = False, **over) -> Cfg:
    bs = (128, 128) if quant == "f8e5m2" else (0, 0)
    if name == "tiny_v3":
        c = Cfg(quant=quant, use_mla=use_mla, block_size=bs)
        if quant == "f8e5m2":  # keep v_head_dim == block_size[0] for F8+MLA (SURVEY 8c pitfalls)
            c.v_head_dim, c.qk_nope_head_dim, c.n_heads = 128, 128, 2
    elif name == "tiny_v2lite":
        c = Cfg(arch="DeepseekV2ForCausalLM", quant=quant, use_mla=False, block_size=bs, q_lora_rank=0,
                n_routed_experts=8, n_active_routed=3, n_shared_experts=2, n_group=1, topk_group=1,
                topk_method="greedy", scoring_func="softmax", norm_topk_prob=False,
                routed_scaling_factor=1.0, moe_intermediate_size=256, hidden_dim=768)
    elif name == "v3":  # true DeepSeek-V3 shapes
        c = Cfg(quant=quant, use_mla=use_mla, block_size=bs, dim=7168, hidden_dim=18432, n_layers=61,
                n_heads=128, vocab_size=129280, max_seq_len=4096, first_k_dense_replace=3,
                n_shared_experts=1, n_routed_experts=256, n_active_routed=8, moe_intermediate_size=2048,
                routed_scaling_factor=2.5, n_group=8, topk_group=4, kv_lora_rank=512, q_lora_rank=1536,
                qk_nope_head_dim=128, qk_rope_head_dim=64, v_head_dim=128)
    elif name == "v2lite":  # DeepSeek-V2-Lite shapes; pad256 variant for K-quants
        kq = quant in ("q2_k", "q3_k")
        c = Cfg(arch="DeepseekV2ForCausalLM", quant=quant, use_mla=False, block_size=bs, dim=2048,
                hidden_dim=11008 if kq else 10944, n_layers=27, n_heads=16, vocab_size=102400,
                max_seq_len=4096, first_k_dense_replace=1, n_shared_experts=2, n_routed_experts=64,
                n_active_routed=6, moe_intermediate_size=1536 if kq else 1408, routed_scaling_factor=1.0,
                n_group=1, topk_group=1, topk_method="greedy", scoring_func="softmax", norm_topk_prob=False,
                kv_lora_rank=512, q_lora_rank=0, qk_nope_head_dim=128, qk_rope_head_dim=64, v_head_dim=128)
    else:
        raise KeyError(name)
    for k, v in over.items():
        setattr(c, k, v)
    return c


# ---------------------------------------------------------------- encoders
def f32_to_f16_bits(x: np.ndarray) -> np.ndarray:
    return x.astype(np.float32).astype(np.float16).view(np.uint16)


def encode_q2k(w: np.ndarray) -> np.ndarray:
    """(rows, n) float32 -> (rows, n/256*84) uint8, layout src/quant.h:41-52."""
    rows, n = w.shape
    nb = n // QK_K
    x = w.reshape(rows * nb, 16, 16).astype(np.float32)
    mn = np.minimum(x.min(axis=2), 0.0)
    mx = x.max(axis=2)
    scale = (mx - mn) / 3.0
    minv = -mn
    d = scale.max(axis=1) / 15.0
    dmin = minv.max(axis=1) / 15.0
    d16 = d.astype(np.float16)
    dmin16 = dmin.astype(np.float16)
    dh = d16.astype(np.float32)
    dminh = dmin16.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        scq = np.where(dh[:, None] > 0, np.rint(scale / dh[:, None]), 0).clip(0, 15).astype(np.uint8)
        mq = np.where(dminh[:, None] > 0, np.rint(minv / dminh[:, None]), 0).clip(0, 15).astype(np.uint8)
        dl = dh[:, None] * scq
        ml = dminh[:, None] * mq
        q = np.where(dl[:, :, None] > 0, np.rint((x + ml[:, :, None]) / dl[:, :, None]), 0).clip(0, 3).astype(np.uint8)
    L = q.reshape(rows * nb, 256)
    out = np.zeros((rows * nb, Q2K_BYTES), dtype=np.uint8)
    out[:, 0:16] = scq | (mq << 4)
    for h in range(2):
        base = L[:, 128 * h:128 * h + 128].reshape(-1, 4, 32)  # [s][l]
        out[:, 16 + 32 * h:16 + 32 * h + 32] = base[:, 0] | (base[:, 1] << 2) | (base[:, 2] << 4) | (base[:, 3] << 6)
    out[:, 80:82] = d16.view(np.uint16).astype("<u2").view(np.uint8).reshape(-1, 2)
    out[:, 82:84] = dmin16.view(np.uint16).astype("<u2").view(np.uint8).reshape(-1, 2)
    return out.reshape(rows, nb * Q2K_BYTES)


def encode_q3k(w: np.ndarray) -> np.ndarray:
    """(rows, n) float32 -> (rows, n/256*110) uint8, layout src/quant.h:70-76."""
    rows, n = w.shape
    nb = n // QK_K
    x = w.reshape(rows * nb, 16, 16).astype(np.float32)
    amax = np.abs(x).max(axis=2)
    scale = amax / 4.0
    d = scale.max(axis=1) / 31.0
    d16 = d.astype(np.float16)
    dh = d16.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        sc = np.where(dh[:, None] > 0, np.rint(scale / dh[:, None]), 0).clip(0, 31).astype(np.int32)
        dl = dh[:, None] * sc
        q = np.where(dl[:, :, None] > 0, np.rint(x / dl[:, :, None]), 0).clip(-4, 3).astype(np.int32) + 4
    l6 = (sc + 32).astype(np.uint8)  # stored 6-bit value, decoded as value - 32
    out = np.zeros((rows * nb, Q3K_BYTES), dtype=np.uint8)
    scales = np.zeros((rows * nb, 12), dtype=np.uint8)
    for j in range(16):  # src/quant.cpp:330-340
        lo = l6[:, j] & 0xF
        if j < 8:
            scales[:, j] |= lo
        else:
            scales[:, j - 8] |= lo << 4
        scales[:, 8 + j % 4] |= (l6[:, j] >> 4) << (2 * (j // 4))
    L = q.reshape(rows * nb, 256).astype(np.uint8)
    hm = np.zeros((rows * nb, 32), dtype=np.uint8)
    for j in range(256):  # src/quant.cpp:361-373: bit (j // 32) of hmask[j % 32]
        hm[:, j % 32] |= ((L[:, j] > 3).astype(np.uint8)) << (j // 32)
    L2 = L & 3
    out[:, 0:32] = hm
    for h in range(2):
        base = L2[:, 128 * h:128 * h + 128].reshape(-1, 4, 32)
        out[:, 32 + 32 * h:32 + 32 * h + 32] = base[:, 0] | (base[:, 1] << 2) | (base[:, 2] << 4) | (base[:, 3] << 6)
    out[:, 96:108] = scales
    out[:, 108:110] = d16.view(np.uint16).astype("<u2").view(np.uint8).reshape(-1, 2)
    return out.reshape(rows, nb * Q3K_BYTES)


def encode_f8_blocks(w: np.ndarray, bs: Tuple[int, int]) -> Tuple[np.ndarray, np.ndarray]:
    """(rows, n) -> (bytes (rows,n) uint8 e5m2, scale (ceil(rows/b0), ceil(n/b1)) f32)."""
    rows, n = w.shape
    b0, b1 = bs
    sr, sc = -(-rows // b0), -(-n // b1)
    scale = np.zeros((sr, sc), dtype=np.float32)
    out = np.zeros((rows, n), dtype=np.uint8)
    for i in range(sr):
        for j in range(sc):
            blk = w[i * b0:(i + 1) * b0, j * b1:(j + 1) * b1].astype(np.float32)
            amax = float(np.abs(blk).max())
            s = amax / 448.0 if amax > 0 else 1.0
            scale[i, j] = s
            out[i * b0:(i + 1) * b0, j * b1:(j + 1) * b1] = (f32_to_f16_bits(blk / s) >> 8).astype(np.uint8)
    return out, scale


@dataclass
class Tens:
    data: np.ndarray          # raw array as stored (uint8 bytes / float16 / float32)
    shape: Tuple[int, ...]    # logical shape (QTensor::shape)
    quant: int                # dsk_quant id
    scale: np.ndarray | None = None  # F8 block scales


def _encode(w: np.ndarray, quant: str, bs) -> Tens:
    """w: (rows, n) or (E, rows, n) float32."""
    qid = QUANT_IDS[quant]
    if quant == "fp32":
        return Tens(np.ascontiguousarray(w, dtype=np.float32), w.shape, qid)
    if quant == "fp16":
        return Tens(np.ascontiguousarray(w.astype(np.float16)), w.shape, qid)
    if quant in ("q2_k", "q3_k"):
        enc = encode_q2k if quant == "q2_k" else encode_q3k
        flat = w.reshape(-1, w.shape[-1])
        data = enc(flat)
        if w.ndim == 3:
            data = data.reshape(w.shape[0], w.shape[1], -1)
        return Tens(np.ascontiguousarray(data), w.shape, qid)
    if quant == "f8e5m2":
        if w.ndim == 3:
            parts = [encode_f8_blocks(w[e], bs) for e in range(w.shape[0])]
            return Tens(np.stack([p[0] for p in parts]), w.shape, qid, np.stack([p[1] for p in parts]))
        b, s = encode_f8_blocks(w, bs)
        return Tens(b, w.shape, qid, s)
    raise KeyError(quant)


# ---------------------------------------------------------------- model synthesis
def synth_model(c: Cfg, seed: int = 1234) -> Dict[str, Tens]:
    """Gaussian weights N(0, 1/fan_in), norms 1+0.1 N(0,1), gate bias 0.1 N(0,1)."""
    rng = np.random.default_rng(seed)
    T: Dict[str, Tens] = {}
    F32 = QUANT_IDS["fp32"]

    def W(name, *shape):
        fan_in = shape[-1]
        w = (rng.standard_normal(shape, dtype=np.float32) / np.sqrt(fan_in)).astype(np.float32)
        T[name + ".weight"] = _encode(w, c.quant, c.block_size)

    def N(name, n):
        T[name + ".weight"] = Tens((1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32), (n,), F32)

    H, hd = c.n_heads, c.head_dim
    # embedding rows ~ N(0,1) so that the residual stream starts at unit scale
    emb = rng.standard_normal((c.vocab_size, c.dim), dtype=np.float32)
    T["model.embed.weight"] = _encode(emb, c.quant, c.block_size)
    N("model.norm", c.dim)
    if not c.tie_embeddings:
        W("model.output", c.vocab_size, c.dim)
    for l in range(c.n_layers):
        p = f"model.layers.{l}."
        N(p + "attn.norm", c.dim)
        N(p + "mlp.norm", c.dim)
        N(p + "attn.kv_a_norm", c.kv_lora_rank)
        if c.q_lora_rank > 0:
            N(p + "attn.q_a_norm", c.q_lora_rank)
            W(p + "attn.wq_a", c.q_lora_rank, c.dim)
        W(p + "attn.wkv_a", c.kv_lora_rank + c.qk_rope_head_dim, c.dim)
        W(p + "attn.wo", c.dim, H * c.v_head_dim)
        if c.use_mla:
            W(p + "attn.wc", H * c.kv_lora_rank, c.q_lora_rank)
            W(p + "attn.wq_rope_b", H * c.qk_rope_head_dim, c.q_lora_rank)
            W(p + "attn.wv_b", H * c.v_head_dim, c.kv_lora_rank)
        else:
            if c.q_lora_rank > 0:
                W(p + "attn.wq_b", H * hd, c.q_lora_rank)
            else:
                W(p + "attn.wq", H * hd, c.dim)
            W(p + "attn.wkv_b", H * (c.qk_nope_head_dim + c.v_head_dim), c.kv_lora_rank)
        if c.n_routed_experts > 0 and l >= c.first_k_dense_replace:
            E, mi = c.n_routed_experts, c.moe_intermediate_size
            W(p + "mlp.w1", E, mi, c.dim)
            W(p + "mlp.w2", E, c.dim, mi)
            W(p + "mlp.w3", E, mi, c.dim)
            if c.n_shared_experts > 0:
                W(p + "shared_mlp.w1", c.n_shared_experts * mi, c.dim)
                W(p + "shared_mlp.w2", c.dim, c.n_shared_experts * mi)
                W(p + "shared_mlp.w3", c.n_shared_experts * mi, c.dim)
            g = (rng.standard_normal((E, c.dim), dtype=np.float32) / np.sqrt(c.dim)).astype(np.float32)
            T[p + "moegate.weight"] = Tens(g, g.shape, F32)
            if c.has_moegate_bias:
                T[p + "moegate.bias"] = Tens((0.1 * rng.standard_normal(E)).astype(np.float32), (E,), F32)
        else:
            W(p + "mlp.w1", c.hidden_dim, c.dim)
            W(p + "mlp.w2", c.dim, c.hidden_dim)
            W(p + "mlp.w3", c.hidden_dim, c.dim)
    return T


def name_to_role(name: str) -> Tuple[int, int]:
    """'.dseek' tensor name -> (dsk_role base id, layer)."""
    base = name.rsplit(".", 1)[0]
    kind = name.rsplit(".", 1)[1]  # weight | scale | bias
    if base == "model.embed":
        return ROLE["EMBED"], -1
    if base == "model.norm":
        return ROLE["FINAL_NORM"], -1
    if base == "model.output":
        return ROLE["OUTPUT"], -1
    parts = base.split(".")
    layer = int(parts[2])
    key = ".".join(parts[3:])
    if key == "moegate" and kind == "bias":
        return ROLE["MOEGATE_BIAS"], layer
    return ROLE[_LAYER_NAMES[key]], layer


def shape4(shape) -> np.ndarray:
    s = np.zeros(4, dtype=np.int32)
    s[:len(shape)] = shape
    return s


def bind_all(T: Dict[str, Tens], bind):
    """bind(role, layer, quant, shape4, ndarray) for every tensor (+ role|SCALE for F8 scales)."""
    for name, t in T.items():
        role, layer = name_to_role(name)
        bind(role, layer, t.quant, shape4(t.shape), t.data)
        if t.scale is not None:
            bind(role + ROLE["SCALE"], layer, QUANT_IDS["fp32"], shape4(t.scale.shape), t.scale)


# ---------------------------------------------------------------- .dseek writer (safetensors layout)
_DT = {np.dtype(np.float32): "F32", np.dtype(np.float16): "F16", np.dtype(np.uint8): "U8"}


def synthetic_vocab(n: int):
    """A vocabulary the reference's Tokenizer (src/tokenizer.cpp) accepts: <s>, </s>, the 256 byte-fallback tokens,
    printable single characters, then two-letter pieces; `n` entries, NUL-separated like convert.py:584 writes them."""
    v = ["<s>", "</s>"] + ["<0x%02X>" % i for i in range(256)]
    v += [chr(i) for i in range(32, 127)]
    letters = "etaoinshrdlucmfwypvbgkqjxz "
    for a in letters:
        for b in letters:
            if len(v) < n:
                v.append(a + b)
    i = 0
    while len(v) < n:
        v.append("tok%d" % i)
        i += 1
    return v[:n]


def write_dseek(dirname: str, c: Cfg, T: Dict[str, Tens], shards: int = 1, tokenizer: bool = False):
    """`shards` files `shard_00k.dseek`: u64 header len | JSON | data (src/codec.cpp:304-331); the metadata goes
    into the first one, tensors are dealt to the files in order (convert.py writes 8 layers per shard).
    tokenizer=True adds `tokenizer.tokens` (U8, NUL-separated: convert.py:584), which the reference's `main` needs."""
    os.makedirs(dirname, exist_ok=True)
    names = list(T.keys())
    per = -(-len(names) // max(1, shards))
    for k in range(max(1, shards)):
        header = {"__metadata__": c.metadata()} if k == 0 else {}
        blobs, off = [], 0

        def add(name, arr: np.ndarray, dtype: str, shape):
            nonlocal off
            b = np.ascontiguousarray(arr).tobytes()
            header[name] = {"dtype": dtype, "shape": [int(s) for s in shape], "data_offsets": [off, off + len(b)]}
            blobs.append(b)
            off += len(b)

        if k == 0 and tokenizer:
            tok = "\0".join(synthetic_vocab(c.vocab_size)).encode("latin-1") + b"\0"
            add("tokenizer.tokens", np.frombuffer(tok, np.uint8), "U8", (len(tok),))
        for name in names[k * per:(k + 1) * per]:
            t = T[name]
            if t.quant == QUANT_IDS["f8e5m2"]:
                add(name, t.data, "F8_E5M2", t.data.shape)
                add(name.rsplit(".", 1)[0] + ".scale", t.scale, "F32", t.scale.shape)
            else:
                add(name, t.data, _DT[t.data.dtype], t.data.shape)
        hj = json.dumps(header).encode()
        with open(os.path.join(dirname, "shard_%03d.dseek" % k), "wb") as f:
            f.write(struct.pack("<Q", len(hj)))
            f.write(hj)
            for b in blobs:
                f.write(b)


def random_block_model(c: Cfg, seed: int = 0, tile_blocks: int = 0) -> Dict[str, "Tens"]:
    """A checkpoint of config `c` whose K-quant tensors are VALID RANDOM BLOCKS (random quant / scale
    bytes, d and dmin chosen so that activations stay finite) instead of encoded gaussians: building it
    costs one random-byte fill, so full-width DeepSeek-V3 layers are practical on the host.  Same recipe
    as the engine's dsk_model_synthesize.  Used by bench.py's CPU baseline and tests/test_fullsize_gpu.py."""
    rng = np.random.default_rng(seed)

    def rand_tensor(shape):  # valid random blocks, like dsk_model_synthesize
        rows = int(np.prod(shape[:-1]))
        n = shape[-1]
        if c.quant == "q2_k":
            nblk = rows * (n // 256)
            # tile_blocks > 0: tensors larger than that many blocks repeat a random pattern (a 12 GB expert stack
            # for CPU timing need not be 12 GB of entropy; every byte is still a distinct address)
            gen = nblk if tile_blocks <= 0 or nblk <= tile_blocks else tile_blocks
            b = rng.integers(0, 256, (gen, 84), dtype=np.uint8)
            d = (rng.uniform(0.5, 1.5, gen) / np.sqrt(n) / 13.9).astype(np.float16)
            b[:, 80:82] = d.view(np.uint8).reshape(-1, 2)
            b[:, 82:84] = (1.5 * d.astype(np.float32)).astype(np.float16).view(np.uint8).reshape(-1, 2)
            if gen < nblk:
                b = np.tile(b, ((nblk + gen - 1) // gen, 1))[:nblk]
            return Tens(b.reshape(*shape[:-1], -1), tuple(shape), QUANT_IDS["q2_k"])
        if c.quant == "q3_k":  # hmask[32] | qs[64] | scales[12] (6-bit, any bytes are valid) | d f16 = 110 B (src/quant.h:70-76)
            nblk = rows * (n // 256)
            gen = nblk if tile_blocks <= 0 or nblk <= tile_blocks else tile_blocks
            b = rng.integers(0, 256, (gen, 110), dtype=np.uint8)
            d = (rng.uniform(0.5, 1.5, gen) / np.sqrt(n) / 43.0).astype(np.float16)  # rms of (q - 4 !h) * (scale - 32) ~ 43
            b[:, 108:110] = d.view(np.uint8).reshape(-1, 2)
            if gen < nblk:
                b = np.tile(b, ((nblk + gen - 1) // gen, 1))[:nblk]
            return Tens(b.reshape(*shape[:-1], -1), tuple(shape), QUANT_IDS["q3_k"])
        w = (rng.standard_normal(shape, dtype=np.float32) / np.sqrt(n)).astype(np.float32)
        return _encode(w, c.quant, c.block_size)

    T = {}
    F32 = QUANT_IDS["fp32"]
    H, hd = c.n_heads, c.head_dim
    T["model.embed.weight"] = rand_tensor((c.vocab_size, c.dim))
    T["model.output.weight"] = rand_tensor((c.vocab_size, c.dim))
    T["model.norm.weight"] = Tens(np.ones(c.dim, np.float32), (c.dim,), F32)
    for l in range(c.n_layers):
        p = f"model.layers.{l}."
        for nm, n in (("attn.norm", c.dim), ("mlp.norm", c.dim), ("attn.kv_a_norm", c.kv_lora_rank)):
            T[p + nm + ".weight"] = Tens(np.ones(n, np.float32), (n,), F32)
        if c.q_lora_rank > 0:
            T[p + "attn.q_a_norm.weight"] = Tens(np.ones(c.q_lora_rank, np.float32), (c.q_lora_rank,), F32)
            T[p + "attn.wq_a.weight"] = rand_tensor((c.q_lora_rank, c.dim))
        T[p + "attn.wkv_a.weight"] = rand_tensor((c.kv_lora_rank + c.qk_rope_head_dim, c.dim))
        T[p + "attn.wo.weight"] = rand_tensor((c.dim, H * c.v_head_dim))
        if c.use_mla:
            T[p + "attn.wc.weight"] = rand_tensor((H * c.kv_lora_rank, c.q_lora_rank))
            T[p + "attn.wq_rope_b.weight"] = rand_tensor((H * c.qk_rope_head_dim, c.q_lora_rank))
            T[p + "attn.wv_b.weight"] = rand_tensor((H * c.v_head_dim, c.kv_lora_rank))
        else:
            if c.q_lora_rank > 0:
                T[p + "attn.wq_b.weight"] = rand_tensor((H * hd, c.q_lora_rank))
            else:
                T[p + "attn.wq.weight"] = rand_tensor((H * hd, c.dim))
            T[p + "attn.wkv_b.weight"] = rand_tensor((H * (c.qk_nope_head_dim + c.v_head_dim), c.kv_lora_rank))
        if l >= c.first_k_dense_replace:
            E, mi = c.n_routed_experts, c.moe_intermediate_size
            T[p + "mlp.w1.weight"] = rand_tensor((E, mi, c.dim))
            T[p + "mlp.w2.weight"] = rand_tensor((E, c.dim, mi))
            T[p + "mlp.w3.weight"] = rand_tensor((E, mi, c.dim))
            if c.n_shared_experts:
                T[p + "shared_mlp.w1.weight"] = rand_tensor((c.n_shared_experts * mi, c.dim))
                T[p + "shared_mlp.w2.weight"] = rand_tensor((c.dim, c.n_shared_experts * mi))
                T[p + "shared_mlp.w3.weight"] = rand_tensor((c.n_shared_experts * mi, c.dim))
            g = (rng.standard_normal((E, c.dim), dtype=np.float32) / np.sqrt(c.dim)).astype(np.float32)
            T[p + "moegate.weight"] = Tens(g, g.shape, F32)
            if c.has_moegate_bias:
                T[p + "moegate.bias"] = Tens(np.zeros(E, np.float32), (E,), F32)
        else:
            T[p + "mlp.w1.weight"] = rand_tensor((c.hidden_dim, c.dim))
            T[p + "mlp.w2.weight"] = rand_tensor((c.dim, c.hidden_dim))
            T[p + "mlp.w3.weight"] = rand_tensor((c.hidden_dim, c.dim))
    return T
