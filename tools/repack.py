"""Offline GPU-layout repack of a `.dseek` checkpoint.

    python tools/repack.py IN_DIR OUT_DIR

The reference's converter (convert.py:296-533) stores K-quant tensors as arrays of 84- / 110-byte blocks (src/quant.h:41-52,
70-76).  The engine streams them as byte PLANES (qs / scales / high bits / d, csrc/dsk_internal.h) and re-lays every
tensor out on the GPU at load time.  This tool persists that layout: for every K-quant tensor `X` of IN_DIR the output holds
`X.qs`, `X.sc`, (`X.hm`,) `X.dm` -- per plane all experts back to back, so an expert-sharded rank reads one contiguous
range per plane -- and a 1-byte marker under the original name; everything else (norms, router, F8 / F16 / F32 tensors,
the tokenizer) is copied verbatim; `__metadata__` gains `gpu_layout = planes-v1`.  dsk_model_load_dseek recognises the
key and copies the planes through the pinned ring straight into their device planes (no DEVICE staging buffer, no repack
kernels: csrc/loader.cpp bind_plane_set).
The reference format stays the interchange format: the reference itself cannot read the repacked directory.
"""
from __future__ import annotations

import json
import os
import struct
import sys

import numpy as np


def read_shard(path):
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(n))
        data0 = 8 + n
    return header, data0


def planes_q2k(blocks: np.ndarray):
    """blocks: (nblk, 84) uint8 = scales[16] | qs[64] | d | dmin  ->  qs (nblk, 64), sc (nblk, 16) in quarter order, dm (nblk, 4)"""
    sc = blocks[:, :16]
    j = np.arange(16)
    h, s, lh = j >> 3, (j >> 1) & 3, j & 1
    out = np.empty_like(sc)
    out[:, (2 * h + lh) * 4 + s] = sc[:, j]  # sc'[4q + s] = scales[8h + 2s + lh], q = 2h + lh (kernels_misc.hip repack_q2k_kernel)
    return {"qs": blocks[:, 16:80], "sc": out, "dm": blocks[:, 80:84]}


def planes_q3k(blocks: np.ndarray):
    """blocks: (nblk, 110) = hmask[32] | qs[64] | scales[12] | d"""
    return {"hm": blocks[:, :32], "qs": blocks[:, 32:96], "sc": blocks[:, 96:108], "dm": blocks[:, 108:110]}


def repack(in_dir: str, out_dir: str):
    os.makedirs(out_dir, exist_ok=True)
    files = sorted(f for f in os.listdir(in_dir))
    meta_quant = None
    for k, fn in enumerate(files):
        header, data0 = read_shard(os.path.join(in_dir, fn))
        md = header.pop("__metadata__", None)
        if k == 0:
            if md is None:
                raise SystemExit("the first file carries no __metadata__")
            meta_quant = md.get("quant")
            if md.get("gpu_layout"):
                raise SystemExit("already repacked")
            md = dict(md, gpu_layout="planes-v1")
        bsz = {"q2_k": 84, "q3_k": 110}.get(meta_quant)
        raw = np.memmap(os.path.join(in_dir, fn), dtype=np.uint8, mode="r")
        out_header = {"__metadata__": md} if md is not None else {}
        blobs, off = [], 0

        def add(name, arr, dtype, shape):
            nonlocal off
            b = np.ascontiguousarray(arr).tobytes()
            out_header[name] = {"dtype": dtype, "shape": [int(x) for x in shape], "data_offsets": [off, off + len(b)]}
            blobs.append(b)
            off += len(b)

        for name, t in header.items():
            a, b = t["data_offsets"]
            data = raw[data0 + a:data0 + b]
            is_kq = bsz is not None and t["dtype"] == "U8" and name.endswith(".weight") and name != "tokenizer.tokens"
            if not is_kq:
                add(name, data, t["dtype"], t["shape"])
                continue
            if data.size % bsz:
                raise SystemExit(f"{name}: {data.size} bytes is not a whole number of {bsz}-byte blocks")
            P = (planes_q2k if bsz == 84 else planes_q3k)(data.reshape(-1, bsz))
            add(name, np.zeros(1, np.uint8), "U8", (1,))  # marker: the loader finds the tensor by its reference name
            for suffix, arr in P.items():  # blocks are (expert, row, block)-major already: a plane is expert-major as is
                add(f"{name}.{suffix}", arr, "U8", (arr.size,))
        hj = json.dumps(out_header).encode()
        with open(os.path.join(out_dir, fn), "wb") as f:
            f.write(struct.pack("<Q", len(hj)))
            f.write(hj)
            for bl in blobs:
                f.write(bl)
    return out_dir


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    repack(sys.argv[1], sys.argv[2])
