"""The C ABI: every symbol declared in include/libzkp_hip.h is exported by the built library; the header
compiles as C; the Python binding lists the same symbols.  No compute calls (no GPU needed)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "libzkp_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zkp_hip_[a-z_0-9]+)\s*\(", src)))


def test_header_is_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "libzkp_hip.h"\nint main(void){return ZKP_HIP_OK;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c), "-o", str(tmp_path / "t.o")])


def test_library_exports_every_declared_symbol():
    from libzkp_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__ as ge
        ge.build_hip()
    syms = declared_symbols()
    assert len(syms) >= 8
    assert sorted(_native.EXPORTS) == syms
    lib = ctypes.CDLL(_native.LIB_PATH)
    for s in syms:
        assert getattr(lib, s) is not None
    nm = subprocess.check_output(["nm", "-D", "--defined-only", _native.LIB_PATH], text=True)
    exported = set(re.findall(r" T (zkp_hip_\w+)", nm))
    assert set(syms) <= exported


def test_rust_bindings_declare_every_symbol():
    """rust/hip_ffi.rs (unbuilt source: no cargo in the image) must declare exactly the header's symbols."""
    rs = open(os.path.join(ROOT, "rust", "hip_ffi.rs")).read()
    assert sorted(set(re.findall(r"pub fn (zkp_hip_[a-z_0-9]+)\s*\(", rs))) == declared_symbols()


def test_cpp_caller_of_every_entry_point_builds_and_fails_loudly_without_gpu():
    """tests/abi/abi_call_all.cpp calls all entry points through the header alone.  Here: it compiles and links against the
    built library, names every declared symbol, and -- on a box without a GPU -- reports the missing device instead of
    computing anything on the CPU.  The GPU tier runs it for real (tests/test_gpu_abi_cpp.py)."""
    import __graft_entry__ as ge
    exe = ge.build_abi_caller()
    src = open(os.path.join(ROOT, "tests", "abi", "abi_call_all.cpp")).read()
    assert sorted(set(re.findall(r"CALLED\((zkp_hip_[a-z_0-9]+)\)", src))) == declared_symbols()
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the real run is in the gpu tier")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device available" in r.stderr


def test_product_path_fails_loudly_without_gpu():
    """No silent CPU fallback: without a device the call errors out (skipped when a GPU is present)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import libzkp_amd
    with pytest.raises((libzkp_amd.ZkpBackendError, libzkp_amd.NativeError)) as ei:
        libzkp_amd.prove_range(5, 0, 10)
    assert "no HIP device" in str(ei.value) or "fallback" in str(ei.value)


def test_product_sources_never_touch_the_oracle():
    for dp, _, fns in os.walk(os.path.join(ROOT, "libzkp_amd")):
        for fn in fns:
            if fn.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libzkp_oracle" not in txt, fn


_BARRIER_CHILD = r"""
import ctypes, resource, sys
L = ctypes.CDLL(sys.argv[1])
L.zkp_hip_last_error.restype = ctypes.c_char_p
u64, u32, vp = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p
L.zkp_hip_prove_threshold_batch.argtypes = [u64, vp, vp, vp, u32, vp, vp, u64, vp, vp]
L.zkp_hip_batch_stage.argtypes = [u64, vp, vp, vp, vp]
dummy = (ctypes.c_uint64 * 8)()
P = ctypes.addressof(dummy)
# 1. absurd sizes are argument errors, checked before anything is sized from them (and before any device is looked for)
out = ctypes.c_void_p()
assert L.zkp_hip_batch_stage(1 << 40, P, P, P, ctypes.addressof(out)) == -3 and b"batch too large" in L.zkp_hip_last_error()
assert L.zkp_hip_prove_threshold_batch(1 << 40, P, P, P, 64, P, P, 762, P, P) == -3
counts = (ctypes.c_uint32 * 2)(0xffffffff, 0xffffffff)
assert L.zkp_hip_prove_threshold_batch(2, P, ctypes.addressof(counts), P, 64, P, P, 762, P, P) == -3 and b"value lists too long" in L.zkp_hip_last_error()
# 2. a std::bad_alloc inside the library (seeds = NULL asks for 32 n bytes of fresh randomness; n is the largest accepted batch, all
#    counts zero so no list value is read) comes back as ZKP_HIP_E_RUNTIME with a message instead of unwinding through the C frame
n = 1 << 22
zeros = (ctypes.c_uint32 * n)()
vm_kb = [int(l.split()[1]) for l in open("/proc/self/status") if l.startswith("VmSize:")][0]
soft, hard = resource.getrlimit(resource.RLIMIT_AS)
resource.setrlimit(resource.RLIMIT_AS, ((vm_kb << 10) + (16 << 20), hard))
rc = L.zkp_hip_prove_threshold_batch(n, P, ctypes.addressof(zeros), P, 64, None, P, 762, P, P)
msg = L.zkp_hip_last_error()
resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
assert rc == -1 and b"out of host memory" in msg, (rc, msg)
# 3. the library is still usable
assert L.zkp_hip_range_proof_bytes(64) == 1478
print("barrier ok")
"""


def test_exception_barrier_at_the_c_abi():
    """per-item status + message, never abort (batch.rs:126-130).  Runs in a child process (it lowers RLIMIT_AS); no compute
    call is made -- every case returns before a device is needed."""
    import sys
    from libzkp_amd import _native
    r = subprocess.run([sys.executable, "-c", _BARRIER_CHILD, _native.LIB_PATH], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "barrier ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
