"""Every entry point of include/libzkp_hip.h called from C++ through the header alone (tests/abi/abi_call_all.cpp): the
boundary a Rust or C++ host binds, exercised without Python in the call path."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_program_calls_every_entry_point():
    import __graft_entry__ as ge
    exe = ge.build_abi_caller()
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "abi_call_all ok: 45 symbols" in r.stdout
