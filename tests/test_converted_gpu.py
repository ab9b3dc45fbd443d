"""VERDICT r1 "missing" #4: a checkpoint produced by the reference's converter, loaded by the HIP loader.

tests/golden/converted_q2k_mla/ and converted_f8e5m2/ were written by the reference's unmodified convert.py from a tiny
HuggingFace-layout model (tools/make_converter_fixture.py): MLA weight absorption wc = W_UK^T W_UQ re-quantised after the
product (convert.py:384-438), per-expert K-quantisation and stacking (:344-362, 488-508), 128 x 128 F8 block scales
(:262-275), all metadata as strings (:123-170), the tokenizer tensor in the first shard.  The recorded logits are the
unmodified reference's on these very files (oracle/_ref/libdskref.so)."""
import os

import numpy as np
import pytest

from tests.util import rel_inf

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["f8e5m2", "q2k_mla"])
def test_converter_made_checkpoint_loads_and_matches_the_reference(ctx, name):
    import dsk
    d = os.path.join(GOLD, "converted_" + name)
    g = np.load(os.path.join(GOLD, f"converted_{name}.npz"))
    M = dsk.Model.from_dseek(ctx, d)
    assert M.load_stats.n_files == 1 and M.load_stats.n_tensors > 30
    toks = [int(t) for t in g["tokens"]]
    e0 = [rel_inf(M.forward(t, 0), g["pos0_logits"][i]) for i, t in enumerate(toks)]
    eseq = [rel_inf(M.forward(t, p), g["seq_logits"][p]) for p, t in enumerate(toks)]
    print(f"\n[converted {name}] pos-0 errors {['%.1e' % e for e in e0]}  sequence {['%.1e' % e for e in eseq]}")
    if name == "f8e5m2":  # float weights: the north star's 1e-3 everywhere
        assert max(e0 + eseq) < 1e-3, (e0, eseq)
    else:
        # W2A8 free-running: tokens without an int8 rounding tie agree to float precision, none is grossly off (a wrong
        # absorbed-weight layout or expert order gives O(1)); the exact per-stage statement is tests/test_teacher_forced_gpu.py
        assert np.median(e0) < 1e-4 and max(e0 + eseq) < 0.1, (e0, eseq)
    M.close()
