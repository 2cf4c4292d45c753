"""ctypes binding of the C ABI in include/libzkp_hip.h (the same symbols the reference's Rust side would bind).

There is deliberately no fallback: if the shared library is missing or no MI355X is present the calls raise.
"""
import atexit
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZKP_HIP_LIB") or os.path.join(_HERE, "lib", "libzkp_hip.so")      # ZKP_HIP_LIB: A/B builds (tools only)

RANGE_PROOF_BYTES = 1478
# symbols declared in include/libzkp_hip.h (checked by tests/test_abi.py)
EXPORTS = (
    "zkp_hip_init", "zkp_hip_shutdown", "zkp_hip_last_error", "zkp_hip_prove_range_batch",
    "zkp_hip_prove_range_batch_device", "zkp_hip_prove_threshold_batch", "zkp_hip_prove_consistency_batch",
    "zkp_hip_consistency_proof_bytes", "zkp_hip_range_proof_bytes", "zkp_hip_threshold_proof_bytes", "zkp_hip_groth16_load_key", "zkp_hip_groth16_key_info", "zkp_hip_groth16_generate_key", "zkp_hip_snark_commit_value_batch",
    "zkp_hip_prove_equality_batch", "zkp_hip_prove_membership_batch", "zkp_hip_improvement_max_bytes", "zkp_hip_prove_improvement_batch",
    "zkp_hip_prove_improvement_batch_device", "zkp_hip_verify_range_batch", "zkp_hip_verify_threshold_batch", "zkp_hip_verify_consistency_batch", "zkp_hip_verify_equality_batch", "zkp_hip_verify_membership_batch",
    "zkp_hip_verify_improvement_batch", "zkp_hip_process_batch", "zkp_hip_profile_enable", "zkp_hip_profile_read", "zkp_hip_set_window_budget", "zkp_hip_set_subbatches", "zkp_hip_set_msm_variant",
    "zkp_hip_init_devices", "zkp_hip_device_count", "zkp_hip_use_device", "zkp_hip_process_batch_bytes", "zkp_hip_batch_stage", "zkp_hip_batch_prove", "zkp_hip_batch_max_bytes",
    "zkp_hip_batch_fetch", "zkp_hip_batch_free", "zkp_hip_profile_read_kernel", "zkp_hip_batch_device_results", "zkp_hip_plan_shards", "zkp_hip_batch_prove_async", "zkp_hip_batch_wait",
)

_lib = None


class Op(ctypes.Structure):
    """zkp_hip_op of include/libzkp_hip.h"""
    _fields_ = [("kind", ctypes.c_uint32), ("count", ctypes.c_uint32), ("a", ctypes.c_uint64), ("b", ctypes.c_uint64), ("c", ctypes.c_uint64),
                ("list_off", ctypes.c_uint64)]


OP_RANGE, OP_EQUALITY, OP_THRESHOLD, OP_MEMBERSHIP, OP_IMPROVEMENT, OP_CONSISTENCY = 1, 2, 3, 4, 5, 6


class NativeError(RuntimeError):
    pass


def _preload_hip_runtime():
    """When PyTorch-ROCm is installed it ships its own libamdhip64.so.7; two HIP runtimes in one process cannot both
    see the GPU.  Loading torch's copy first (same soname) makes libzkp_hip.so and torch share one runtime whichever
    is imported first.  Without torch the system ROCm runtime is used."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                "libzkp_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                "libzkp_amd has no CPU fallback")
        _preload_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        u64, u32, i32, vp = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p
        L.zkp_hip_init.argtypes = [ctypes.c_int]
        L.zkp_hip_init.restype = ctypes.c_int
        L.zkp_hip_shutdown.restype = None
        L.zkp_hip_last_error.restype = ctypes.c_char_p
        L.zkp_hip_prove_range_batch.argtypes = [u64, vp, vp, vp, u32, vp, vp, u64, vp, vp]
        L.zkp_hip_prove_range_batch.restype = ctypes.c_int
        L.zkp_hip_prove_range_batch_device.argtypes = [u64, vp, vp, vp, u32, vp, vp, u64, vp, vp, vp, ctypes.POINTER(ctypes.c_int)]
        L.zkp_hip_prove_range_batch_device.restype = ctypes.c_int
        L.zkp_hip_prove_threshold_batch.argtypes = [u64, vp, vp, vp, u32, vp, vp, u64, vp, vp]
        L.zkp_hip_prove_threshold_batch.restype = ctypes.c_int
        L.zkp_hip_prove_consistency_batch.argtypes = [u64, vp, vp, vp, vp, u64, vp, vp]
        L.zkp_hip_prove_consistency_batch.restype = ctypes.c_int
        L.zkp_hip_consistency_proof_bytes.argtypes = [u32]
        L.zkp_hip_consistency_proof_bytes.restype = u64
        for f in (L.zkp_hip_range_proof_bytes, L.zkp_hip_threshold_proof_bytes):
            f.argtypes = [u32]
            f.restype = u64
        L.zkp_hip_groth16_load_key.argtypes = [ctypes.c_int, vp, u64]
        L.zkp_hip_groth16_load_key.restype = ctypes.c_int
        L.zkp_hip_groth16_key_info.argtypes = [ctypes.c_int, ctypes.POINTER(u32), ctypes.POINTER(u32), ctypes.POINTER(u64)]
        L.zkp_hip_groth16_key_info.restype = ctypes.c_int
        L.zkp_hip_groth16_generate_key.argtypes = [ctypes.c_int, vp, vp, u64, ctypes.POINTER(u64), vp, u64, ctypes.POINTER(u64)]
        L.zkp_hip_groth16_generate_key.restype = ctypes.c_int
        L.zkp_hip_snark_commit_value_batch.argtypes = [u64, vp, vp]
        L.zkp_hip_snark_commit_value_batch.restype = ctypes.c_int
        L.zkp_hip_prove_equality_batch.argtypes = [u64, vp, vp, vp, vp, u64, vp, vp]
        L.zkp_hip_prove_equality_batch.restype = ctypes.c_int
        L.zkp_hip_prove_membership_batch.argtypes = [u64, vp, vp, vp, vp, vp, u64, vp, vp]
        L.zkp_hip_prove_membership_batch.restype = ctypes.c_int
        L.zkp_hip_improvement_max_bytes.argtypes = []
        L.zkp_hip_improvement_max_bytes.restype = u32
        L.zkp_hip_prove_improvement_batch.argtypes = [u64, vp, vp, vp, u64, vp, vp]
        L.zkp_hip_prove_improvement_batch.restype = ctypes.c_int
        L.zkp_hip_prove_improvement_batch_device.argtypes = [u64, vp, vp, vp, u64, vp, vp]
        L.zkp_hip_prove_improvement_batch_device.restype = ctypes.c_int
        L.zkp_hip_verify_range_batch.argtypes = [u64, vp, u64, vp, vp, vp, vp]
        L.zkp_hip_verify_range_batch.restype = ctypes.c_int
        L.zkp_hip_verify_threshold_batch.argtypes = [u64, vp, u64, vp, vp, vp]
        L.zkp_hip_verify_threshold_batch.restype = ctypes.c_int
        L.zkp_hip_verify_consistency_batch.argtypes = [u64, vp, u64, vp, vp]
        L.zkp_hip_verify_consistency_batch.restype = ctypes.c_int
        L.zkp_hip_verify_equality_batch.argtypes = [u64, vp, u64, vp, vp]
        L.zkp_hip_verify_equality_batch.restype = ctypes.c_int
        L.zkp_hip_verify_membership_batch.argtypes = [u64, vp, u64, vp, vp]
        L.zkp_hip_verify_membership_batch.restype = ctypes.c_int
        L.zkp_hip_verify_improvement_batch.argtypes = [u64, vp, u64, vp, vp, vp]
        L.zkp_hip_verify_improvement_batch.restype = ctypes.c_int
        L.zkp_hip_process_batch.argtypes = [u64, vp, vp, vp, vp, u64, vp, vp]
        L.zkp_hip_process_batch.restype = ctypes.c_int
        L.zkp_hip_profile_enable.argtypes = [ctypes.c_int]
        L.zkp_hip_profile_enable.restype = None
        L.zkp_hip_profile_read.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.c_int]
        L.zkp_hip_profile_read.restype = ctypes.c_int
        L.zkp_hip_set_window_budget.argtypes = [u32]
        L.zkp_hip_set_window_budget.restype = None
        L.zkp_hip_set_subbatches.argtypes = [u32]
        L.zkp_hip_set_subbatches.restype = None
        L.zkp_hip_set_msm_variant.argtypes = [u32]
        L.zkp_hip_set_msm_variant.restype = None
        L.zkp_hip_init_devices.argtypes = [u32, ctypes.POINTER(ctypes.c_int)]
        L.zkp_hip_init_devices.restype = ctypes.c_int
        L.zkp_hip_device_count.argtypes = []
        L.zkp_hip_device_count.restype = ctypes.c_int
        L.zkp_hip_use_device.argtypes = [ctypes.c_int]
        L.zkp_hip_use_device.restype = ctypes.c_int
        L.zkp_hip_process_batch_bytes.argtypes = [u64, vp, ctypes.POINTER(u64)]
        L.zkp_hip_process_batch_bytes.restype = ctypes.c_int
        L.zkp_hip_batch_stage.argtypes = [u64, vp, vp, vp, ctypes.POINTER(vp)]
        L.zkp_hip_batch_stage.restype = ctypes.c_int
        for f in (L.zkp_hip_batch_prove, L.zkp_hip_batch_prove_async, L.zkp_hip_batch_wait):
            f.argtypes = [vp]
            f.restype = ctypes.c_int
        L.zkp_hip_batch_max_bytes.argtypes = [vp]
        L.zkp_hip_batch_max_bytes.restype = u64
        L.zkp_hip_batch_fetch.argtypes = [vp, vp, u64, vp, vp]
        L.zkp_hip_batch_fetch.restype = ctypes.c_int
        L.zkp_hip_batch_device_results.argtypes = [vp, u32, vp, u64, vp, ctypes.POINTER(u64), vp]
        L.zkp_hip_batch_device_results.restype = ctypes.c_int
        L.zkp_hip_plan_shards.argtypes = [u64, vp, u32, vp]
        L.zkp_hip_plan_shards.restype = ctypes.c_int
        L.zkp_hip_batch_free.argtypes = [vp]
        L.zkp_hip_batch_free.restype = None
        L.zkp_hip_profile_read_kernel.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.c_int]
        L.zkp_hip_profile_read_kernel.restype = ctypes.c_int
        # Release every device object while the interpreter and the HIP runtime are still fully alive (the library also
        # registers its own C atexit hook at the first initialisation; both are idempotent).  DESIGN.md "exit-time teardown".
        atexit.register(L.zkp_hip_shutdown)
        _lib = L
    return _lib


def init_devices(devices):
    """zkp_hip_init_devices: shard k of the library on HIP device devices[k] (one process drives them all)."""
    arr = (ctypes.c_int * len(devices))(*devices)
    return check(lib().zkp_hip_init_devices(len(devices), arr), "zkp_hip_init_devices")


def last_error():
    return lib().zkp_hip_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc < 0:
        raise NativeError("%s failed (%d): %s" % (what, rc, last_error()))
    return rc
