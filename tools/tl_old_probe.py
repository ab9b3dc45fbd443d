import ctypes as C, os, sys
import numpy as np
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deepseek.cpp_amd"))
import dsk
from tools import synth
c = synth.preset("v3", "q2_k", False, n_layers=8, max_seq_len=64)
ctx = dsk.Ctx(0)
M = dsk.Model(ctx, c, None, synth_seed=0, options={"timeline": 1})
for pos in range(6): M.forward(17 + pos, pos)
buf = np.zeros((256, 8), np.uint64)
f = dsk.lib().dsk_model_get_moe_timeline; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
dsk.check(f(M.h, buf.ctypes.data, 256))
t0 = buf[:,0].min()
lo0 = (buf[:,0] & np.uint64(0xffffffff)).astype(np.int64)
for name, col in (("wave 2", 7),):
    raw = buf[:, col]
    req = (raw >> np.uint64(32)).astype(np.int64); mac = (raw & np.uint64(0xffffffff)).astype(np.int64)
    ent = (buf[:,0] - t0).astype(np.float64)/100
    r = ((req - lo0) & 0xffffffff)/100.0 + ent; m = ((mac - lo0) & 0xffffffff)/100.0 + ent
    print(f"old kernel {name}: first group requested min/med/max {r.min():.2f} {np.median(r):.2f} {r.max():.2f}; multiplied {m.min():.2f} {np.median(m):.2f} {m.max():.2f}")
a_done = (buf[:,2]-t0).astype(np.float64)/100
print(f"phase A done {a_done.min():.2f} {np.median(a_done):.2f} {a_done.max():.2f}")
