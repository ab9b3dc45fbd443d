"""The N > 1 path on CPU: expert ownership (host arithmetic of the engine, no GPU needed) and the
exchange step of an expert-sharded MoE layer run by TWO processes over gloo.

The engine's multi-GPU scheme (csrc/forward.cpp ffn): rank r owns experts
[base, base + count) of every routed stack; per MoE layer every rank zero-fills the per-slot output
buffer, computes only the slots whose expert it owns, one sum all-reduce over (K, dim) floats, then
the k-ordered combine  x += w_k * out_k  (src/infer.cpp:874-877).  Every slot is non-zero on exactly
one rank, so the sum is exact and the result must be BIT-identical to the single-process one.  Here
the per-slot expert FFN is computed by the oracle (the checker, CPU) instead of the HIP kernels; the
ownership map comes from the product library (dsk_expert_shard).
"""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))


def shard(n_experts, world, rank):
    import dsk
    base, count = C.c_int(), C.c_int()
    dsk.check(dsk.lib().dsk_expert_shard(n_experts, world, rank, C.byref(base), C.byref(count)))
    return base.value, count.value


def owner(n_experts, world, expert):
    import dsk
    r = C.c_int()
    f = dsk.lib().dsk_expert_owner
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    dsk.check(f(n_experts, world, expert, C.byref(r)))
    return r.value


@pytest.mark.parametrize("E", [1, 6, 8, 64, 160, 256])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_expert_shards_partition_the_stack(E, world):
    seen = []
    for r in range(world):
        base, count = shard(E, world, r)
        assert 0 <= count <= -(-E // world)
        seen.extend(range(base, base + count))
    assert seen == list(range(E))  # contiguous, disjoint, complete, in rank order
    # ownership as the kernels evaluate it (csrc/kernels_gemv.hip resolve(): le = e - base, 0 <= le < count)
    per = -(-E // world)
    for e in range(E):
        o = e // per
        base, count = shard(E, world, o)
        assert base <= e < base + count
        assert owner(E, world, e) == o  # the rank the all-gather form of the exchange reads slot outputs from


def test_expert_shard_rejects_bad_arguments():
    import dsk
    b, c = C.c_int(), C.c_int()
    assert dsk.lib().dsk_expert_shard(8, 0, 0, C.byref(b), C.byref(c)) == -1
    assert dsk.lib().dsk_expert_shard(8, 2, 2, C.byref(b), C.byref(c)) == -1
    assert dsk.lib().dsk_expert_shard(8, 2, 0, None, C.byref(c)) == -1


def tp_rows(rows, unit, world, rank):
    import dsk
    f = dsk.lib().dsk_tp_rows
    f.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] * 2
    r0, n = C.c_int(), C.c_int()
    dsk.check(f(rows, unit, world, rank, C.byref(r0), C.byref(n)))
    return r0.value, n.value


# every GEMV of a DeepSeek-V3 token (rows, unit): unit = 256 where the consumer quantises the vector in Q8_K blocks (a block
# must come from ONE rank to be quantised before the gather), the head size for per-head projections, 1 for the classifier
V3_GEMVS = [(1536, 256), (576, 64), (128 * 192, 192), (128 * 256, 256), (7168, 256), (18432, 256), (2048, 256), (256, 1), (129280, 1)]


@pytest.mark.parametrize("rows,unit", V3_GEMVS)
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_tensor_parallel_row_ranges_partition_every_gemv(rows, unit, world):
    got, sizes = [], []
    for r in range(world):
        r0, n = tp_rows(rows, unit, world, r)
        assert r0 % unit == 0 and n % unit == 0
        got.extend(range(r0, r0 + n))
        sizes.append(n)
    assert got == list(range(rows))                 # contiguous, disjoint, complete, in rank order
    assert max(sizes) - min(sizes) <= unit          # balanced to one unit
    import dsk
    h0, hn = C.c_int(), C.c_int()
    f = dsk.lib().dsk_tp_heads
    f.argtypes = [C.c_int] * 3 + [C.POINTER(C.c_int)] * 2
    heads = []
    for r in range(world):
        dsk.check(f(128, world, r, C.byref(h0), C.byref(hn)))
        heads.extend(range(h0.value, h0.value + hn.value))
    assert heads == list(range(128))
    r0, n = C.c_int(), C.c_int()
    g = dsk.lib().dsk_tp_rows
    assert g(100, 256, 2, 0, C.byref(r0), C.byref(n)) == -1     # rows not a multiple of the unit
    assert g(512, 256, 2, 2, C.byref(r0), C.byref(n)) == -1     # rank out of range


# ---------------------------------------------------------------------------------------------
# two ranks over gloo
# ---------------------------------------------------------------------------------------------
DIM, INTER, E, K = 512, 256, 8, 3
Q2K = 3


def _layer(seed):
    """A tiny MoE layer: Q2_K expert stacks (valid reference blocks), normed input, fixed routing."""
    from tools import synth
    rng = np.random.default_rng(seed)
    w1 = np.stack([synth.encode_q2k(rng.standard_normal((INTER, DIM)).astype(np.float32) / np.sqrt(DIM)) for _ in range(E)])
    w3 = np.stack([synth.encode_q2k(rng.standard_normal((INTER, DIM)).astype(np.float32) / np.sqrt(DIM)) for _ in range(E)])
    w2 = np.stack([synth.encode_q2k(rng.standard_normal((DIM, INTER)).astype(np.float32) / np.sqrt(INTER)) for _ in range(E)])
    xb = rng.standard_normal(DIM).astype(np.float32)
    x = rng.standard_normal(DIM).astype(np.float32)
    experts = np.array([6, 1, 3], np.int32)  # slots in k order, owners: rank 1, 0, 0 for world = 2
    weights = np.array([0.5, 0.3, 0.2], np.float32)
    return w1, w2, w3, xb, x, experts, weights


def _slot(orc, w1, w2, w3, xb, e):
    """One routed slot: W2_e (silu(W1_e xb) * W3_e xb), src/infer.cpp:853-873."""
    h1 = orc.gemv_expert(Q2K, w1, int(e), INTER, DIM, xb)
    h3 = orc.gemv_expert(Q2K, w3, int(e), INTER, DIM, xb)
    hb = (h1 / (np.float32(1.0) + np.exp(-h1, dtype=np.float32))).astype(np.float32) * h3
    return orc.gemv_expert(Q2K, w2, int(e), DIM, INTER, hb)


def _combine(x, outs, weights):
    x = x.copy()
    for k in range(K):  # k order, one f32 fma-free multiply-add per element like the reference's loop
        x = (x + outs[k] * weights[k]).astype(np.float32)
    return x


def _worker(rank, world, port, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from oracle import orc as orcmod
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = orcmod.Oracle()
        w1, w2, w3, xb, x, experts, weights = _layer(seed)
        base, count = shard(E, world, rank)
        eout = np.zeros((K, DIM), np.float32)  # zero-filled, like the hipMemsetAsync before the W2 launch
        mine = 0
        for k in range(K):
            if base <= experts[k] < base + count:
                eout[k] = _slot(orc, w1[:, :, :], w2, w3, xb, experts[k])
                mine += 1
        t = torch.from_numpy(eout.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out = _combine(x, t.numpy(), weights)
        # the all-gather form (option "exchange_allgather", csrc/forward.cpp ffn): every rank contributes its K slot rows as they
        # are - rows of experts it does not own hold junk here, NaN, to prove nobody reads them - and slot k is taken from the
        # copy of the rank that owns expert k (dsk_expert_owner, what moe_combine_gathered_kernel evaluates)
        mine_rows = np.full((K, DIM), np.nan, np.float32)
        for k in range(K):
            if base <= experts[k] < base + count:
                mine_rows[k] = eout[k]
        parts = [torch.zeros(K, DIM) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(mine_rows))
        picked = np.stack([parts[owner(E, world, int(experts[k]))].numpy()[k] for k in range(K)])
        out_g = _combine(x, picked, weights)
        # the timing protocol of bench.py: barrier, then the MAX over ranks of a per-rank duration
        dist.barrier()
        dt = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        q.put((rank, mine, out.tobytes(), float(dt.item()), out_g.tobytes()))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_two_rank_expert_sharded_layer_is_bit_identical_to_one_rank(oracle):
    import torch.multiprocessing as mp
    seed = 11
    w1, w2, w3, xb, x, experts, weights = _layer(seed)
    ref = _combine(x, np.stack([_slot(oracle, w1, w2, w3, xb, e) for e in experts]), weights)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [2, 1]  # slots computed per rank: experts 1, 3 on rank 0; 6 on rank 1
    for rank, _, blob, tmax, blob_g in res:
        out = np.frombuffer(blob, np.float32)
        assert np.array_equal(out, ref), f"rank {rank}: sharded result differs from the single-rank result"
        assert np.array_equal(np.frombuffer(blob_g, np.float32), ref), f"rank {rank}: all-gather form differs"
        assert tmax == 2.0  # every rank sees the slowest rank's time


def _tp_worker(rank, world, port, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from oracle import orc as orcmod
    from tools import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = orcmod.Oracle()
        rng = np.random.default_rng(seed)
        rows, n = 1024, 512
        w = synth.encode_q2k(rng.standard_normal((rows, n)).astype(np.float32) / np.sqrt(n))
        x = rng.standard_normal(n).astype(np.float32)
        r0, cnt = tp_rows(rows, 256, world, rank)
        part = orc.gemv(Q2K, np.ascontiguousarray(w[r0:r0 + cnt]), cnt, n, x)   # this rank's rows only
        parts = [torch.zeros(tp_rows(rows, 256, world, r)[1], dtype=torch.float32) for r in range(world)]
        dist.all_gather(parts, torch.from_numpy(part.copy()))                       # equal ranges here: a plain all-gather
        q.put((rank, torch.cat(parts).numpy().tobytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_row_split_gemv_all_gather_is_bit_identical(oracle):
    """DESIGN.md 4.4 (row f-4): a replicated GEMV split by OUTPUT rows over the ranks and re-assembled with an all-gather of
    the token's vector - what the north star words - is bit-identical to the one-rank GEMV (rows are independent)."""
    import torch.multiprocessing as mp
    from tools import synth
    seed = 5
    rng = np.random.default_rng(seed)
    w = synth.encode_q2k(rng.standard_normal((1024, 512)).astype(np.float32) / np.sqrt(512))
    x = rng.standard_normal(512).astype(np.float32)
    ref = oracle.gemv(Q2K, w, 1024, 512, x)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, port, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, blob in res:
        assert np.array_equal(np.frombuffer(blob, np.float32), ref), rank
