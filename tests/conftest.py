"""pytest configuration: markers, import paths, shared fixtures.

  -m "not gpu": oracle vs the reference's golden vectors, host logic, C-ABI surface (runs on CPU)
  -m gpu      : parity tests proper -- every call goes through the C ABI of libdsk_hip.so
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepseek.cpp_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import orc
    if not os.path.exists(orc.ORC_SO):
        orc.build(ref=os.path.isdir("/root/reference"))
    return orc.Oracle()


@pytest.fixture(scope="session")
def ref():
    """The unmodified reference (oracle/_ref/libdskref.so): built here, prebuilt on the GPU box."""
    from oracle import orc
    if not os.path.exists(orc.REF_SO):
        if os.path.isdir("/root/reference"):
            orc.build(ref=True)
        else:
            pytest.skip("oracle/_ref/libdskref.so not present")
    try:
        r = orc.Ref()
    except OSError as e:  # e.g. a host CPU without AVX2/F16C
        pytest.skip(f"reference library not loadable: {e}")
    r.set_threads(4)
    return r


@pytest.fixture(scope="session")
def ops_gold():
    return dict(np.load(os.path.join(GOLD, "ops.npz")))


@pytest.fixture(scope="session")
def ctx():
    import dsk
    c = dsk.Ctx(0)
    yield c
    c.close()
