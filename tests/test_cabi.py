"""The drop-in boundary on CPU: the shared library loads, exports every symbol include/dsk.h
declares, its POD config matches the ctypes mirror, and without a GPU it fails loudly instead of
falling back to anything."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "dsk.h")


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dsk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import dsk
    L = dsk.lib()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/dsk.h but not exported by libdsk_hip.so"
    assert L.dsk_abi_version() == 1


def test_config_struct_layout_matches_header():
    import dsk
    prog = '#include <stdio.h>\n#include "dsk.h"\nint main(){printf("%zu %zu %zu", sizeof(dsk_config), ' \
           '__builtin_offsetof(dsk_config, block_size), __builtin_offsetof(dsk_config, rs_original_max_position_embeddings));}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        size, off_bs, off_rs = map(int, subprocess.check_output([os.path.join(d, "t")]).split())
    assert C.sizeof(dsk.DskConfig) == size
    assert dsk.DskConfig.block_size.offset == off_bs
    assert dsk.DskConfig.rs_original_max_position_embeddings.offset == off_rs
    from oracle import orc
    assert C.sizeof(orc.DskConfig) == size
    prog = '#include <stdio.h>\n#include "dsk.h"\nint main(){printf("%zu %zu %zu", sizeof(dsk_load_stats), ' \
           '__builtin_offsetof(dsk_load_stats, seconds), __builtin_offsetof(dsk_load_stats, n_tensors));}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        size, off_s, off_n = map(int, subprocess.check_output([os.path.join(d, "t")]).split())
    assert (C.sizeof(dsk.LoadStats), dsk.LoadStats.seconds.offset, dsk.LoadStats.n_tensors.offset) == (size, off_s, off_n)


def test_no_gpu_means_loud_failure_not_fallback():
    import dsk
    L = dsk.lib()
    h = C.c_void_p()
    r = L.dsk_ctx_create(0, C.byref(h))
    if r == 0:  # a GPU is visible (GPU box): nothing to assert here
        L.dsk_ctx_destroy(h)
        pytest.skip("GPU present")
    assert r < 0 and len(L.dsk_last_error()) > 0
    with pytest.raises(dsk.DskError):
        dsk.Ctx(0)


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under deepseek.cpp_amd/ may import, link or call it."""
    pkg = os.path.join(ROOT, "deepseek.cpp_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("liborc", "libdskref", "from oracle", "import oracle", "orc_forward", "dsk_oracle.h"):
                    assert needle not in txt, (f, needle)
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libdsk_hip.so")]).decode()
    assert "liborc" not in out and "libdskref" not in out
