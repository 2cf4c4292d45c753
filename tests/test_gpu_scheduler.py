"""The mixed-batch scheduler behind zkp_hip_process_batch (batch.rs:110-140,262-283): bytes equal to the per-variant entry
points op for op, the staged entry points, the capacity query, and the multi-shard path -- one process driving several
shards -- run on ONE GPU by registering the same HIP device twice."""
import ctypes
import os

import numpy as np
import pytest

from libzkp_amd import workloads as wl
from util import P

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib(devices=None):
    from libzkp_amd import _native
    import libzkp_amd.api as api
    L = _native.lib()
    if devices is not None:
        L.zkp_hip_shutdown()
        _native.init_devices(devices)
    else:
        _native.check(L.zkp_hip_init(0), "zkp_hip_init")
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        blob = open(os.path.join(GOLD, name), "rb").read()
        assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0, _native.last_error()
    with api._snark_lock:
        api._keys_loaded[0] = api._keys_loaded[1] = True
        api._reinstall.clear()
    return L


def _run(L, ops, lists, seeds, cap=None):
    from libzkp_amd import _native
    n = len(ops)
    cap = wl.max_output_bytes(ops) if cap is None else cap
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64)
    st = np.zeros(n, dtype=np.int32)
    rc = L.zkp_hip_process_batch(n, P(ops), P(lists), P(seeds), P(out), cap, P(off), P(st))
    assert rc >= 0, _native.last_error()
    return rc, [out[int(off[i]):int(off[i + 1])].tobytes() for i in range(n)], st, off


def _per_variant(L, ops, lists, seeds):
    """The same ops through the per-variant entry points (the composition the scheduler must reproduce byte for byte)."""
    n = len(ops)
    sd = seeds.reshape(n, 32)
    res = [None] * n
    k = ops["kind"]
    ix = np.nonzero(k == wl.OP_RANGE)[0]
    if len(ix):
        m = len(ix); o = np.zeros((m, 1478), dtype=np.uint8); ln = np.zeros(m, dtype=np.uint32); st = np.zeros(m, dtype=np.int32)
        a, b, c, s = ops["a"][ix].copy(), ops["b"][ix].copy(), ops["c"][ix].copy(), np.ascontiguousarray(sd[ix])
        assert L.zkp_hip_prove_range_batch(m, P(a), P(b), P(c), 64, P(s), P(o), 1478, P(ln), P(st)) >= 0
        for j, i in enumerate(ix):
            res[i] = o[j, :ln[j]].tobytes()
    ix = np.nonzero(k == wl.OP_EQUALITY)[0]
    if len(ix):
        m = len(ix); o = np.zeros((m, 298), dtype=np.uint8); ln = np.zeros(m, dtype=np.uint32); st = np.zeros(m, dtype=np.int32)
        a, b, s = ops["a"][ix].copy(), ops["b"][ix].copy(), np.ascontiguousarray(sd[ix])
        assert L.zkp_hip_prove_equality_batch(m, P(a), P(b), P(s), P(o), 298, P(ln), P(st)) >= 0
        for j, i in enumerate(ix):
            res[i] = o[j, :ln[j]].tobytes()
    ix = np.nonzero(k == wl.OP_MEMBERSHIP)[0]
    if len(ix):
        m = len(ix); stride = 10 + 4 + 8 * 64 + 256 + 32
        o = np.zeros((m, stride), dtype=np.uint8); ln = np.zeros(m, dtype=np.uint32); st = np.zeros(m, dtype=np.int32)
        a, cnt, s = ops["a"][ix].copy(), ops["count"][ix].copy(), np.ascontiguousarray(sd[ix])
        flat = np.concatenate([lists[int(ops["list_off"][i]):int(ops["list_off"][i]) + int(ops["count"][i])] for i in ix] + [np.zeros(1, dtype=np.uint64)])
        assert L.zkp_hip_prove_membership_batch(m, P(a), P(flat), P(cnt), P(s), P(o), stride, P(ln), P(st)) >= 0
        for j, i in enumerate(ix):
            res[i] = o[j, :ln[j]].tobytes()
    ix = np.nonzero(k == wl.OP_IMPROVEMENT)[0]
    if len(ix):
        m = len(ix); o = np.zeros((m, 3527), dtype=np.uint8); ln = np.zeros(m, dtype=np.uint32); st = np.zeros(m, dtype=np.int32)
        a, b = ops["a"][ix].copy(), ops["b"][ix].copy()
        assert L.zkp_hip_prove_improvement_batch(m, P(a), P(b), P(o), 3527, P(ln), P(st)) >= 0
        for j, i in enumerate(ix):
            res[i] = o[j, :ln[j]].tobytes()
    ix = np.nonzero(k == wl.OP_THRESHOLD)[0]
    if len(ix):
        m = len(ix); o = np.zeros((m, 762), dtype=np.uint8); ln = np.zeros(m, dtype=np.uint32); st = np.zeros(m, dtype=np.int32)
        th, cnt, s = ops["a"][ix].copy(), ops["count"][ix].copy(), np.ascontiguousarray(sd[ix])
        flat = np.concatenate([lists[int(ops["list_off"][i]):int(ops["list_off"][i]) + int(ops["count"][i])] for i in ix] + [np.zeros(1, dtype=np.uint64)])
        assert L.zkp_hip_prove_threshold_batch(m, P(flat), P(cnt), P(th), 64, P(s), P(o), 762, P(ln), P(st)) >= 0
        for j, i in enumerate(ix):
            res[i] = o[j, :ln[j]].tobytes()
    ix = np.nonzero(k == wl.OP_CONSISTENCY)[0]
    if len(ix):
        m = len(ix); cnt = ops["count"][ix].copy(); stride = int(max(L.zkp_hip_consistency_proof_bytes(int(c)) for c in cnt))
        o = np.zeros((m, stride), dtype=np.uint8); ln = np.zeros(m, dtype=np.uint32); st = np.zeros(m, dtype=np.int32)
        s = np.ascontiguousarray(sd[ix])
        flat = np.concatenate([lists[int(ops["list_off"][i]):int(ops["list_off"][i]) + int(ops["count"][i])] for i in ix] + [np.zeros(1, dtype=np.uint64)])
        assert L.zkp_hip_prove_consistency_batch(m, P(flat), P(cnt), P(s), P(o), stride, P(ln), P(st)) >= 0
        for j, i in enumerate(ix):
            res[i] = o[j, :ln[j]].tobytes()
    return res


def _six_kind_batch(n, seed):
    """All six variants interleaved, some ops invalid (they must fail per item and leave no bytes)."""
    rng = np.random.default_rng(seed)
    ops = np.zeros(n, dtype=wl.OP_DTYPE)
    lists = []
    for i in range(n):
        k = i % 6
        o = ops[i]
        if k == 0:
            o["kind"] = wl.OP_RANGE; o["a"] = rng.integers(0, 2**32); o["c"] = 2**32
            if i % 30 == 0:
                o["a"] = 2**33                                        # out of range
        elif k == 1:
            o["kind"] = wl.OP_EQUALITY; o["a"] = rng.integers(0, 2**63); o["b"] = o["a"]
            if i % 42 == 1:
                o["b"] = o["a"] + np.uint64(1)                        # not equal
        elif k == 2:
            s = rng.choice(2**32, int(rng.integers(1, 20)), replace=False).astype(np.uint64)
            o["kind"] = wl.OP_MEMBERSHIP; o["count"] = len(s); o["list_off"] = len(lists); o["a"] = s[i % len(s)]
            if i % 54 == 2:
                o["a"] = 2**40                                        # not in the set
            lists.extend(int(x) for x in s)
        elif k == 3:
            o["kind"] = wl.OP_IMPROVEMENT; o["a"] = rng.integers(0, 2**63); o["b"] = o["a"] + np.uint64(1 + int(rng.integers(0, 2**32)))
            if i % 66 == 3:
                o["b"] = o["a"]                                       # no improvement
        elif k == 4:
            v = rng.integers(0, 2**40, int(rng.integers(1, 5))).astype(np.uint64)
            o["kind"] = wl.OP_THRESHOLD; o["count"] = len(v); o["list_off"] = len(lists); o["a"] = int(v.sum()) // 2
            if i % 78 == 4:
                o["a"] = int(v.sum()) + 1                             # threshold above the sum
            lists.extend(int(x) for x in v)
        else:
            v = np.sort(rng.integers(0, 2**40, int(rng.integers(1, 5))).astype(np.uint64))
            o["kind"] = wl.OP_CONSISTENCY; o["count"] = len(v); o["list_off"] = len(lists)
            if i % 90 == 5 and len(v) > 1:
                v = v[::-1].copy(); v[0] += np.uint64(1)              # decreasing
            lists.extend(int(x) for x in v)
    return ops, np.array(lists + [0], dtype=np.uint64), wl.op_seeds(seed, n)


def test_scheduler_equals_per_variant_calls_on_all_six_kinds():
    L = _lib()
    ops, lists, seeds = _six_kind_batch(180, 11)
    rc, got, st, off = _run(L, ops, lists, seeds)
    want = _per_variant(L, ops, lists, seeds)
    assert rc == 1                                                   # some ops are invalid by construction
    bad = np.nonzero(st)[0]
    assert len(bad) >= 5 and all(len(got[i]) == 0 for i in bad)
    assert all(st[i] == 1 for i in bad)                              # ZkpError::InvalidInput
    for i in range(len(ops)):
        assert got[i] == want[i], (i, int(ops["kind"][i]))
    assert int(off[-1]) == sum(len(p) for p in want)


def test_c5_mix_bytes_and_order():
    L = _lib()
    ops, lists, seeds = wl.mixed_ops(512, 5)
    rc, got, st, off = _run(L, ops, lists, seeds)
    assert rc == 0 and not st.any()
    want = _per_variant(L, ops, lists, seeds)
    assert got == want
    assert [p[1] for p in got[:8]] == [1, 2, 4, 5, 1, 2, 4, 5]      # envelope scheme ids follow the caller's order


def test_capacity_query_and_small_buffer():
    from libzkp_amd import _native
    L = _lib()
    ops, lists, seeds = wl.mixed_ops(64, 9)
    need = ctypes.c_uint64(0)
    assert L.zkp_hip_process_batch_bytes(len(ops), P(ops), ctypes.byref(need)) == 0
    assert need.value == wl.max_output_bytes(ops)
    rc, got, st, off = _run(L, ops, lists, seeds, cap=need.value)
    assert rc == 0 and int(off[-1]) <= need.value
    out = np.zeros(16, dtype=np.uint8); off2 = np.zeros(len(ops) + 1, dtype=np.uint64); st2 = np.zeros(len(ops), dtype=np.int32)
    assert L.zkp_hip_process_batch(len(ops), P(ops), P(lists), P(seeds), P(out), 16, P(off2), P(st2)) == -3
    assert int(off2[-1]) == int(off[-1]) and "too small" in _native.last_error()


def test_staged_batch_reproves_identically():
    from libzkp_amd import _native
    L = _lib()
    ops, lists, seeds = wl.mixed_ops(128, 21)
    h = ctypes.c_void_p()
    assert L.zkp_hip_batch_stage(len(ops), P(ops), P(lists), P(seeds), ctypes.byref(h)) == 0, _native.last_error()
    cap = int(L.zkp_hip_batch_max_bytes(h))
    assert cap == wl.max_output_bytes(ops)
    outs = []
    for _ in range(2):
        assert L.zkp_hip_batch_prove(h) == 0, _native.last_error()
        out = np.zeros(cap, dtype=np.uint8); off = np.zeros(len(ops) + 1, dtype=np.uint64); st = np.zeros(len(ops), dtype=np.int32)
        assert L.zkp_hip_batch_fetch(h, P(out), cap, P(off), P(st)) == 0
        outs.append(out[:int(off[-1])].tobytes())
    L.zkp_hip_batch_free(h)
    assert outs[0] == outs[1]
    rc, got, st, off = _run(L, ops, lists, seeds)
    assert b"".join(got) == outs[0]


def test_two_shards_on_one_gpu_reproduce_single_shard_bytes():
    """One process, two shards (the same physical GPU registered twice): every variant's bucket is cut into two contiguous
    slices, each shard is driven by its own host thread, results land in the caller's one buffer in op order."""
    from libzkp_amd import _native
    L = _lib()
    mixed = wl.mixed_ops(256, 5)
    six = _six_kind_batch(96, 3)
    single = [_run(L, *b) for b in (mixed, six)]
    L = _lib(devices=[0, 0])
    try:
        assert L.zkp_hip_device_count() == 2
        for b, (rc1, got1, st1, off1) in zip((mixed, six), single):
            rc2, got2, st2, off2 = _run(L, *b)
            assert rc2 == rc1 and (st2 == st1).all() and (off2 == off1).all()
            assert got2 == got1
        # the per-variant entry points follow the calling thread's shard selection
        assert L.zkp_hip_use_device(1) == 0
        v = np.array([7], dtype=np.uint64); lo = np.zeros(1, dtype=np.uint64); hi = np.array([100], dtype=np.uint64)
        o = np.zeros((1, 1478), dtype=np.uint8); ln = np.zeros(1, dtype=np.uint32); st = np.zeros(1, dtype=np.int32)
        sd = np.arange(32, dtype=np.uint8)
        assert L.zkp_hip_prove_range_batch(1, P(v), P(lo), P(hi), 64, P(sd), P(o), 1478, P(ln), P(st)) == 0
        o1 = o.copy()
        assert L.zkp_hip_use_device(0) == 0
        assert L.zkp_hip_prove_range_batch(1, P(v), P(lo), P(hi), 64, P(sd), P(o), 1478, P(ln), P(st)) == 0
        assert (o == o1).all()
        assert L.zkp_hip_use_device(2) == -3
    finally:
        L.zkp_hip_shutdown()
        _lib()


def test_two_batches_in_flight_on_one_shard():
    """zkp_hip_batch_prove_async / _wait: two staged batches on the shard's two lanes of streams, launched before either is
    waited for, several rounds; bytes equal the blocking calls (workspaces of a lane are reused stream-ordered)."""
    from libzkp_amd import _native
    L = _lib()
    batches = [wl.mixed_ops(160, 31), _six_kind_batch(96, 7), wl.mixed_ops(64, 32)]
    want = [_run(L, *b) for b in batches]
    hs = []
    for b in batches:
        h = ctypes.c_void_p()
        assert L.zkp_hip_batch_stage(len(b[0]), P(b[0]), P(b[1]), P(b[2]), ctypes.byref(h)) == 0, _native.last_error()
        hs.append(h)
    for rnd in range(3):
        for h in hs:
            assert L.zkp_hip_batch_prove_async(h) == 0, _native.last_error()
        for h in reversed(hs):
            assert L.zkp_hip_batch_wait(h) == 0
    for h, b, (rc, got, st, off) in zip(hs, batches, want):
        n = len(b[0]); cap = int(L.zkp_hip_batch_max_bytes(h))
        out = np.zeros(cap, dtype=np.uint8); o2 = np.zeros(n + 1, dtype=np.uint64); s2 = np.zeros(n, dtype=np.int32)
        assert L.zkp_hip_batch_fetch(h, P(out), cap, P(o2), P(s2)) == rc
        assert (o2 == off).all() and (s2 == st).all() and out[:int(o2[-1])].tobytes() == b"".join(got)
        L.zkp_hip_batch_free(h)


def test_fewer_ops_than_shards_and_empty_staged_batch():
    """A shard that owns no op still reports out_off[0] = 0 (two shards, one op), and an empty staged batch proves and
    fetches as nothing."""
    from libzkp_amd import _native
    L = _lib()
    one = tuple(x[:1].copy() if i == 0 else x for i, x in enumerate(wl.mixed_ops(4, 77)))         # one range op
    ref = _run(L, one[0], one[1], one[2][:32].copy())
    h = ctypes.c_void_p()
    assert L.zkp_hip_batch_stage(0, None, None, None, ctypes.byref(h)) == 0, _native.last_error()
    assert L.zkp_hip_batch_prove(h) == 0, _native.last_error()
    off = np.full(1, 99, dtype=np.uint64); out = np.zeros(8, dtype=np.uint8)
    assert L.zkp_hip_batch_fetch(h, P(out), 8, P(off), None) == 0, _native.last_error()
    assert int(off[0]) == 0
    L.zkp_hip_batch_free(h)
    L = _lib(devices=[0, 0])
    try:
        for nops in (1, 3):
            ops, lists, seeds = wl.mixed_ops(4, 77)
            ops, seeds = ops[:nops].copy(), seeds[:32 * nops].copy()
            rc, got, st, off = _run(L, ops, lists, seeds)
            assert rc == 0 and not st.any() and int(off[-1]) == sum(len(g) for g in got)
            assert got[0] == ref[1][0]
    finally:
        L.zkp_hip_shutdown()
        _lib()


def test_per_variant_groth16_call_while_a_batch_is_in_flight():
    """zkp_hip_prove_equality_batch / _membership_batch use lane 0's Groth16 workspace; a batch launched with
    zkp_hip_batch_prove_async on that lane may still be running: the second user waits for the first (stream-ordered)."""
    from libzkp_amd import _native
    L = _lib()
    b = wl.mixed_ops(512, 41)
    want = _run(L, *b)
    e_ops, e_lists, e_seeds = wl.equality_ops(300, 43)
    want_e = _per_variant(L, e_ops, e_lists, e_seeds)
    h = ctypes.c_void_p()
    assert L.zkp_hip_batch_stage(len(b[0]), P(b[0]), P(b[1]), P(b[2]), ctypes.byref(h)) == 0, _native.last_error()
    for rnd in range(2):
        assert L.zkp_hip_batch_prove_async(h) == 0, _native.last_error()
        got_e = _per_variant(L, e_ops, e_lists, e_seeds)             # while the batch runs
        assert L.zkp_hip_batch_wait(h) == 0
        assert got_e == want_e
        n = len(b[0]); cap = int(L.zkp_hip_batch_max_bytes(h))
        out = np.zeros(cap, dtype=np.uint8); o2 = np.zeros(n + 1, dtype=np.uint64); s2 = np.zeros(n, dtype=np.int32)
        assert L.zkp_hip_batch_fetch(h, P(out), cap, P(o2), P(s2)) == 0
        assert out[:int(o2[-1])].tobytes() == b"".join(want[1])
    L.zkp_hip_batch_free(h)


def test_missing_key_fails_before_anything_is_enqueued_and_a_retry_works():
    from libzkp_amd import _native
    L = _lib()
    b = wl.mixed_ops(64, 51)
    want = _run(L, *b)
    L.zkp_hip_shutdown()
    _native.check(L.zkp_hip_init(0), "zkp_hip_init")                 # no keys in this life of the shard
    n = len(b[0]); cap = wl.max_output_bytes(b[0])
    out = np.zeros(cap, dtype=np.uint8); off = np.zeros(n + 1, dtype=np.uint64); st = np.zeros(n, dtype=np.int32)
    assert L.zkp_hip_process_batch(n, P(b[0]), P(b[1]), P(b[2]), P(out), cap, P(off), P(st)) == -3
    assert "no proving key" in _native.last_error()
    L = _lib()
    got = _run(L, *b)
    assert got[1] == want[1]


def test_cu_partition_knob_and_launch_trace_leave_the_bytes_alone(tmp_path):
    """ZKP_HIP_BP_CUS (the CU partition of mixed batches: masked streams, grids sized per partition; off by default) and ZKP_HIP_TRACE
    (event records around every launch) change how a batch is scheduled, never what it proves.  Both are read once per process, so the
    variant runs in a child process; the trace file must hold one line per proved batch with the kernels of every variant."""
    import json
    import subprocess
    import sys
    from libzkp_amd import _native
    L = _lib()
    ops, lists, seeds = wl.mixed_ops(1024, 61)                        # 256 of each variant: enough for the partition to engage
    want = _run(L, ops, lists, seeds)
    trace = str(tmp_path / "trace.jsonl")
    code = r'''
import ctypes, hashlib, os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from libzkp_amd import _native, workloads as wl
from util import P
L = _native.lib(); _native.check(L.zkp_hip_init(0), "init")
for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    blob = open(os.path.join(%r, name), "rb").read(); assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0
ops, lists, seeds = wl.mixed_ops(1024, 61)
cap = wl.max_output_bytes(ops); out = np.zeros(cap, dtype=np.uint8); off = np.zeros(1025, dtype=np.uint64); st = np.zeros(1024, dtype=np.int32)
for _ in range(2):
    assert L.zkp_hip_process_batch(1024, P(ops), P(lists), P(seeds), P(out), cap, P(off), P(st)) == 0
print(hashlib.sha256(out[:int(off[-1])].tobytes()).hexdigest())
L.zkp_hip_shutdown()
''' % (ROOT_DIR, ROOT_DIR, GOLD)
    import hashlib
    digest = hashlib.sha256(b"".join(want[1])).hexdigest()
    for extra in ({"ZKP_HIP_BP_CUS": "8"}, {"ZKP_HIP_TRACE": trace}):
        env = dict(os.environ, ZKP_HIP_G16_WBITS="11", **extra)       # small tables: a second process on the GPU
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        assert p.stdout.strip().splitlines()[-1] == digest, extra
    lines = [json.loads(x) for x in open(trace) if x.strip()]
    assert len(lines) == 2
    names = {r[0] for r in lines[-1]}
    assert {"k_msm_gather<EdGather>", "k_msm_gather<G1Msm>", "k_msm_gather<G2Msm>", "k_g16_qap", "k_stark_prove", "k_transcript_round"} <= names
    assert all(r[3] >= r[2] >= 0 for r in lines[-1])
