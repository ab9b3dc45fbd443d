"""The RCCL exchange of the expert-sharded path, EXECUTED.

The builder's boxes have one GPU, so until now ncclAllReduce on the engine's stream had never run anywhere.  A one-rank
communicator (dsk_comm_init(uid, 0, 1)) plus the model option "force_exchange" runs the whole sharded code path on one
GPU - two-launch experts with the shared expert as their ninth task, zero-fill of absent slots, ncclAllReduce(sum) over
the K x dim slot buffer on the engine's non-blocking stream, the separate k-ordered combine launch - and the result
must be BIT-identical to the unsharded model (a sum over one rank is the identity; the slot outputs use the same
summation trees).  Eager first, then from a captured hipGraph ("graph_with_comm").
"""
import numpy as np
import pytest

from tools import synth

pytestmark = pytest.mark.gpu


def _comm_ctx():
    import dsk
    c = dsk.Ctx(0)
    c.comm_init(c.comm_unique_id(), 0, 1)
    return c


@pytest.mark.parametrize("gather", [0, 1], ids=["allreduce", "allgather"])
@pytest.mark.parametrize("width", ["tiny", "v3"])
def test_forced_exchange_runs_rccl_on_the_engine_stream_and_keeps_the_bits(ctx, width, gather):
    """gather = 1: option "exchange_allgather" - ncclAllGather of the ranks' slot rows, the combine reads every slot from its
    owner's copy (one rank here: owner 0); same bits, eager and captured"""
    import dsk
    if width == "tiny":
        c = synth.preset("tiny_v3", "q2_k", False)
        T, seed = synth.synth_model(c, seed=17), None
    else:  # DeepSeek-V3 width, 64 experts, 1 dense + 2 MoE blocks
        c = synth.preset("v3", "q2_k", True, n_layers=3, first_k_dense_replace=1, n_routed_experts=64, max_seq_len=64)
        T, seed = None, 4
    n_moe = c.n_layers - c.first_k_dense_replace
    A = dsk.Model(ctx, c, T, synth_seed=seed)
    cc = _comm_ctx()
    B = dsk.Model(cc, c, T, synth_seed=seed, options={"force_exchange": 1, "graph_with_comm": 0, "exchange_allgather": gather})
    assert B.info("fused_moe_layers") == 0
    toks = [5, 9, 700, 3, 44]
    for pos, t in enumerate(toks):
        la, lb = A.forward(t % c.vocab_size, pos), B.forward(t % c.vocab_size, pos)
        assert np.array_equal(la, lb), pos
        assert np.array_equal(A.routing()[0], B.routing()[0]) and np.array_equal(A.slot_outputs(), B.slot_outputs())
    assert B.info("exchange_calls") == n_moe * len(toks)   # one collective per MoE layer and token, all eager
    assert B.info("graph_captured") == 0
    B.close()
    # the same step captured into a hipGraph (RCCL is initialised by the first, eager, token)
    G = dsk.Model(cc, c, T, synth_seed=seed, options={"force_exchange": 1, "exchange_allgather": gather})  # graph_with_comm is the default
    try:
        for pos, t in enumerate(toks):
            lg = G.forward(t % c.vocab_size, pos)
            assert np.array_equal(lg, A.forward(t % c.vocab_size, pos)), pos
        captured = G.info("graph_captured")
    except dsk.DskError as e:  # recorded, not hidden: DESIGN.md 4.4 states which it is
        pytest.xfail(f"the collective inside hipStreamBeginCapture failed on this ROCm / RCCL: {e}")
    assert captured >= 1
    # replays are bit-stable
    ref = G.forward(7, len(toks)).copy()
    for _ in range(50):
        assert np.array_equal(G.forward_nocopy(7, len(toks)), ref)
    G.close()
    A.close()
    cc.close()


def test_bench_under_torchrun_runs_the_rank_plumbing(tmp_path):
    """the path a driver would launch - `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` -
    reads RANK / LOCAL_RANK / WORLD_SIZE, initialises torch.distributed over RCCL, broadcasts the ncclUniqueId, builds the engine's
    communicator and (option force_exchange) runs the expert-sharded step with its RCCL exchange on a 4-block model; and
    `bench.py --gpus 2` WITHOUT a launcher must start its ranks itself or fail - never print a 1-GPU line."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29731",
           os.path.join(root, "bench.py"), "--gpus", "1", "--layers", "4", "--steps", "6", "--warmup", "3", "--ctx", "64", "--no-cpu-baseline", "--no-extras",
           "--opt", "force_exchange=1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, stdin=subprocess.DEVNULL)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["engine"]["exchange_calls"] > 0, d["engine"]
    # --gpus 2 on a one-GPU box without a launcher: the self-spawned ranks cannot both get a GPU -> non-zero exit, no JSON line
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--layers", "2", "--steps", "2", "--warmup", "1", "--ctx", "64",
                         "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, env=env, stdin=subprocess.DEVNULL)
    import torch
    if torch.cuda.device_count() < 2:
        assert r2.returncode != 0 and not any(l.startswith('{"metric"') for l in r2.stdout.splitlines())
