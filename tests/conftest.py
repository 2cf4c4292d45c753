import ctypes
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _golden(name):
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_ristretto():
    return _golden("ristretto_libsodium.json")


@pytest.fixture(scope="session")
def golden_bp():
    return _golden("bulletproofs_oracle_vectors.json")


@pytest.fixture(scope="session")
def oracle_c():
    """The C oracle (checker).  Built on demand from oracle/c; never used by the product path."""
    import __graft_entry__ as ge
    ge.build_oracle()
    lib = ctypes.CDLL(ge.ORACLE_LIB)
    lib.zkp_oracle_init()
    return lib


@pytest.fixture(scope="session")
def emul():
    """Host builds of the device step functions (tests/emul)."""
    import __graft_entry__ as ge
    ge.build_emul()
    d = os.path.join(ge.EMUL_DIR, "_build")
    return ctypes.CDLL(os.path.join(d, "libemul_math.so")), ctypes.CDLL(os.path.join(d, "libemul_bp.so"))
