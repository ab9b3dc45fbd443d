"""Handles of the C ABI may be dropped in any order (include/dsk.h: a context destroyed while models are alive is freed by
the last dsk_model_destroy), device memory comes back, and the library ignores the environment."""
import ctypes as C
import subprocess
import sys
import os

import numpy as np
import pytest

from tools import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_bytes():
    import torch
    return torch.cuda.mem_get_info(0)[0]


def test_context_and_models_destroyed_in_either_order():
    """dsk_model_destroy must hand its context back (live_models), so that a context destroyed FIRST is freed by
    the last model, and a context destroyed LAST is freed at once; pinned staging buffers, the stream and the models' HBM
    are released either way (no growth over repeated cycles)."""
    import dsk
    L = dsk.lib()
    c = synth.preset("tiny_v3", "q2_k", False)
    T = synth.synth_model(c, seed=3)
    ref = None
    free0 = None
    for cycle in range(6):
        ctx = dsk.Ctx(0)
        assert ctx.live_models() == 0
        A = dsk.Model(ctx, c, T)
        B = dsk.Model(ctx, c, T, options={"fuse_moe": 0})
        assert ctx.live_models() == 2
        la = A.forward(5, 0)
        ref = la if ref is None else ref
        assert np.array_equal(la, ref) and np.array_equal(B.forward(5, 0), ref)
        if cycle % 2 == 0:   # models first, context last
            A.close()
            assert ctx.live_models() == 1
            B.close()
            assert ctx.live_models() == 0
            ctx.close()
        else:                # context first: marked, then freed by the last model
            h = ctx.h.value
            ctx.close()
            assert L.dsk_ctx_live_models(C.c_void_p(h)) == 2   # still alive (closing)
            with pytest.raises(dsk.DskError):                 # ... but closed for new models
                D = dsk.Ctx.__new__(dsk.Ctx); D.h = C.c_void_p(h); D.rank, D.world = 0, 1
                dsk.Model(D, c, T)
            A.close()
            B.close()                                          # frees the context
        if cycle == 1:
            free0 = _free_bytes()
    assert _free_bytes() >= free0 - (64 << 20), (free0, _free_bytes())   # nothing accumulates over 4 more cycles


def test_library_ignores_the_environment():
    """a drop-in .so must not change kernels because of a stray variable in the host's environment.
    With every former knob set, a fresh process computes the same bits, still fuses the expert launch, and has no timeline."""
    code = r"""
import sys, os, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'deepseek.cpp_amd'))
import dsk
from tools import synth
c = synth.preset('tiny_v3', 'q2_k', False, n_shared_experts=0)   # (fuses its expert launch: see test_fused_moe_gpu.py)
T = synth.synth_model(c, seed=3)
ctx = dsk.Ctx(0); M = dsk.Model(ctx, c, T)
out = [M.forward(t, p).copy() for p, t in enumerate((5, 9, 2))]
try:
    M.timeline(4); tl = 1
except dsk.DskError:
    tl = 0
print('FUSED', M.info('fused_moe_layers'), 'TL', tl, 'SUM', repr(float(np.sum(np.stack(out).astype(np.float64)))))
""" % (ROOT, ROOT)
    knobs = dict(DSK_NO_FUSE_MOE="1", DSK_NO_FUSE_SHARED="1", DSK_NO_KVWRITE_RIDE="1", DSK_ATT_Q8_IN_WO="1", DSK_RIDER_FILL="2",
                 DSK_TIMELINE="1", DSK_MOE_TIMELINE="1", DSK_NO_COMPACT="1", DSK_MHA_SPLIT_MIN="16", DSK_MLA_FLASH_MIN="32",
                 DSK_NO_HINT="1", DSK_SMALL_NW16="0", DSK_FORCE_NW="4")
    env_clean = {k: v for k, v in os.environ.items() if not k.startswith("DSK_")}
    a = subprocess.check_output([sys.executable, "-c", code], env=env_clean).decode().strip().splitlines()[-1]
    b = subprocess.check_output([sys.executable, "-c", code], env=dict(env_clean, **knobs)).decode().strip().splitlines()[-1]
    assert a == b and a.startswith("FUSED") and " TL 0 " in a, (a, b)
    assert int(a.split()[1]) > 0
