#!/usr/bin/env python3
"""Headline benchmark: process_batch of a 4096-op MIXED batch per GPU (BASELINE.json's metric).

Workload (SURVEY.md 8d, C5's mix at the metric's 4096-op size): op i is prove_range(v, 0, 2^32) / prove_equality(a, a) /
prove_membership(value, 16-element set) / prove_improvement(old, new) for i mod 4 = 0 / 1 / 2 / 3, seed 5 (+ rank), per-proof
seeds SHA-256(seed || i).  One "step" = zkp_hip_batch_prove on that batch: the whole scheduler of zkp_hip_process_batch
(batch.rs:110-140 replacement) with the batch already staged in HBM and the packed proofs + offsets left in HBM.  The
same batch through zkp_hip_process_batch with host buffers in and out (staging, H2D, D2H included) is reported beside it as
`host_buffers` -- it is never `value`.

Multi-GPU: one process per GPU (the driver launches `python -m torch.distributed.run ... bench.py --gpus N`; a bare
`python bench.py --gpus N` launches those children itself before anything touches the GPU).
  * the contract's line (`value`, "scaling": "weak"): every rank proves its own 4096-op batch (the ops are independent, there is no
    data-path exchange inside the proving) and the packed proofs of all ranks are gathered to every rank with one RCCL all_gather per
    step inside the timed region; the gather of step k overlaps the proving of step k + 1 and all have landed before the clock stops.
  * `strong_scaling_c5` (extra key, N > 1 or --force-dist): BASELINE config 5 itself -- ONE seeded 16 384-op mixed batch, rank r stages
    the ops zkp_hip_plan_shards gives it (per-variant contiguous slices: every GPU runs the same kernel mix), proves them, and the
    all_gather of the packed proofs is inside the clock, nothing overlapped: value = 16 384 x steps / wall.
  * `in_library_shards` (extra key, N = 1 when the process sees several GPUs): the same 16 384-op batch through ONE process driving
    every visible GPU (zkp_hip_init_devices + zkp_hip_process_batch), the form the reference's single Rust process would use.

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline       the dominant kernel of the mixed batch, k_msm_gather<G1Msm> (Groth16 key-point MSMs), against the HBM roof
  roofline_valu  the same kernel against the VALU integer roof that actually binds it
  cpu_baseline   oracle/c's process_batch port (OpenMP over ops, like rayon) on a bounded sample of the same ops.
and at N = 1, beside the contract's keys: host_buffers, two_batches_in_flight, other_configs_staged (C2 / C3 / C4 on their own),
batch_size_sweep (the mixed batch at 512 ... 16 384 ops) with predicted_strong_scaling (what one 16 384-op batch over N GPUs can reach, from
this one GPU's step times), g16_table_radix / g16_table_bytes (which key tables produced `value`) and verification_c_abi (4096 Groth16 equality envelopes and 4096 range envelopes through the C ABI, and 65 536 equality envelopes through the one-pairing-check path).
The timed region runs with the library's per-launch event profiling OFF; the launch durations behind `roofline` come from a second
pass of the same K steps with it on (`profiled_pass` holds that pass's step time: the cost of the instrumentation is visible).
"""
import argparse
import csv
import ctypes
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 4096
C5_BATCH = 16384
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md
MAD_ISSUE_T = 35.157                     # T v_mad_u64_u32 lane-ops/s, isolated issue-rate measurement (profiles/r02_fe_microbench.json)
MADS_PER_G1_MADD = 7 * 162 + 2 * 126 + 243      # g1_mmadd9 (XYZZ, nine 29-bit limbs): 7 products, 2 squarings, 1 fused double product (bn254_fq9.h)
G1_BARE_LOOP_GADDS = 16.27               # G additions/s of the bare addition loop at 3 waves/SIMD, no loads, cold and sustained over 4 s alike (tools/g1_add_rate.hip, profiles/r04_g1_add_rate.jsonl; round 2 on another box: 16.99)
MADS_PER_ED_MADD = 7 * 100               # mixed addition with an affine-Niels entry: 7 GF(2^255-19) products of 100 mads
ED_KERNEL = "k_msm_gather<EdGather>"      # round 4: HBM-resident radix-2^16 generator tables (libzkp_amd/csrc/edg.h); rounds 1-3: k_msm_dma<EdMsm>
ALGO_BYTES = {"range": 24 + 32 + 1478, "equality": 16 + 32 + 298, "membership16": 8 + 128 + 32 + 430}
def _latest(name):
    """profiles/rNN_<name> of the latest round that has one"""
    for r in range(9, 0, -1):
        path = os.path.join(ROOT, "profiles", "r%02d_%s" % (r, name))
        if os.path.exists(path):
            return path
    return os.path.join(ROOT, "profiles", "r04_" + name)


ALONE_CSV = _latest("kernel_alone.csv")      # per-kernel durations with every dispatch serialised (counter pass)
TRAFFIC_JSON = _latest("traffic.json")
XGMI_LINK_GBS = 153.0                    # per link and direction (MI355X_MICROARCH.md); a ring all_gather is bound by one link


def csrc_sha16():
    """Identifies the kernel sources a profile was taken on: sha256 over libzkp_amd/csrc (names + contents), first 16 hex digits."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "libzkp_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _relaunch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks as children of a process that has not touched the GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def cpu_baseline(sample, threads):
    """The mixed batch on the host cores: oracle/c's port of process_batch (one OpenMP task per op), on the first `sample`
    ops of the same workload.  kind = "port": C restatements of upstream's algorithms, not the Rust crates themselves."""
    import numpy as np
    from libzkp_amd import workloads as wl
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libzkp_oracle.so"))
    orc.zkp_oracle_init()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    u64 = ctypes.c_uint64
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        pk = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
        assert orc.zkp_oracle_g16_load_key(kind, pk, u64(len(pk))) == 0
    ops, lists, seeds = wl.mixed_ops(BATCH, 5)
    ops, seeds = ops[:sample].copy(), seeds[:32 * sample].copy()
    cap = wl.max_output_bytes(ops)
    out = np.zeros(cap, dtype=np.uint8); off = np.zeros(sample + 1, dtype=np.uint64); st = np.zeros(sample, dtype=np.int32)
    warm = min(sample, 4 * threads)
    orc.zkp_oracle_process_batch(u64(warm), P(ops), P(lists), P(seeds), P(out), u64(cap), P(off), P(st), threads)      # thread pool, page faults
    t0 = time.perf_counter()
    rc = orc.zkp_oracle_process_batch(u64(sample), P(ops), P(lists), P(seeds), P(out), u64(cap), P(off), P(st), threads)
    dt = time.perf_counter() - t0
    assert rc == 0
    # BASELINE configs[0]: the reference's own harness shape, benchmark_proof_generation("range", 100) = 100 x prove_range(50, 0, 100), one thread
    k = 100
    v1, mn1, mx1 = np.full(k, 50, dtype=np.uint64), np.zeros(k, dtype=np.uint64), np.full(k, 100, dtype=np.uint64)
    sd1 = wl.op_seeds(7, k)
    out1, len1, st1 = np.zeros((k, 1478), dtype=np.uint8), np.zeros(k, dtype=np.uint32), np.zeros(k, dtype=np.int32)
    t1 = time.perf_counter()
    assert orc.zkp_oracle_prove_range_batch(u64(k), P(v1), P(mn1), P(mx1), 64, P(sd1), P(out1), u64(1478), P(len1), P(st1), 1) == 0
    dt1 = time.perf_counter() - t1
    return {"value": sample / dt, "unit": "proofs/s", "cores": threads, "kind": "port",
            "c1_single_thread": {"workload": "100 x prove_range(50, 0, 100), 1 thread (BASELINE configs[0])", "ms_per_proof": dt1 / k * 1e3},
            "sample": "first %d ops of the same 4096-op mixed batch (%d each of range / equality / membership(16) / improvement) through oracle/c's "
                      "process_batch port, OpenMP %d threads, %.1f s wall = %.0f core-seconds" % (sample, sample // 4, threads, dt, dt * threads)}


def alone_durations():
    """Average duration of a kernel with every dispatch serialised (profiles/r03_kernel_alone.csv, written by tools/kernel_alone.py from the
    counter pass of tools/profile_round.sh): what the kernel takes when it has the GPU to itself."""
    if not os.path.exists(ALONE_CSV):
        return {}
    with open(ALONE_CSV) as f:
        return {r["kernel"]: {"avg_ms": float(r["avg_ms"]), "launches_per_step": float(r["launches_per_step"])} for r in csv.DictReader(f)}


def alone_build():
    """The build the serialised pass was taken on (sidecar written by tools/kernel_alone.py), so that a stale file is visible in the line."""
    meta = ALONE_CSV.replace(".csv", ".meta.json")
    return json.load(open(meta)).get("csrc_sha16") if os.path.exists(meta) else None


def in_library_child(devices, nt, steps):
    """The `in_library_shards` leg: ONE process drives every listed GPU (zkp_hip_init_devices, one host worker thread per shard) through a
    16 384-op C5 batch.  Prints one JSON object."""
    import numpy as np
    from libzkp_amd import _native, workloads as wl
    L = _native.lib()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _native.init_devices(devices)
    t0 = time.perf_counter()
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        blob = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
        _native.check(L.zkp_hip_groth16_load_key(kind, blob, len(blob)), "zkp_hip_groth16_load_key")
    t_keys = time.perf_counter() - t0

    def timed(f, reps):
        f()
        ts = []
        for _ in range(reps):
            t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
        return statistics.median(ts)
    ops5, lists5, seeds5 = wl.mixed_ops(nt, 5)
    h5 = ctypes.c_void_p()
    _native.check(L.zkp_hip_batch_stage(nt, P(ops5), P(lists5), P(seeds5), ctypes.byref(h5)), "stage (in-library shards)")
    t_st = timed(lambda: _native.check(L.zkp_hip_batch_prove(h5), "prove"), max(3, steps // 4))
    cap5 = int(L.zkp_hip_batch_max_bytes(h5))
    b5 = np.zeros(cap5, dtype=np.uint8); o5 = np.zeros(nt + 1, dtype=np.uint64); c5 = np.zeros(nt, dtype=np.int32)
    assert _native.check(L.zkp_hip_batch_fetch(h5, P(b5), cap5, P(o5), P(c5)), "fetch") == 0 and not c5.any()
    L.zkp_hip_batch_free(h5)
    t_hb5 = timed(lambda: _native.check(L.zkp_hip_process_batch(nt, P(ops5), P(lists5), P(seeds5), P(b5), cap5, P(o5), P(c5)), "process_batch"), 3)
    L.zkp_hip_shutdown()
    print(json.dumps({"devices": devices, "ops_in_the_one_batch": nt, "key_load_s_all_shards": t_keys,
                      "staged": {"value": nt / t_st, "unit": "proofs/s", "ms_per_batch": t_st * 1e3},
                      "host_buffers": {"value": nt / t_hb5, "unit": "proofs/s", "ms_per_batch": t_hb5 * 1e3},
                      "note": "one process, zkp_hip_init_devices(%d shards), one host worker thread per shard; the %d-op C5 batch cut into per-variant contiguous "
                              "slices; run in a child process after the main measurement; not the contract's value" % (len(devices), nt)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the legs reported beside the contract's line (host buffers, C2/C3/C4 on their own, strong scaling, in-library shards)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the driver's runs) or gloo (rehearsing N > 1 on a box with fewer GPUs)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and run the gather path even at world size 1 (tests/test_gpu_rccl.py)")
    ap.add_argument("--c5-batch", type=int, default=C5_BATCH, help="ops of the strong-scaling leg's one batch")
    ap.add_argument("--cpu-sample", type=int, default=1024)
    ap.add_argument("--child-in-library", default="", help=argparse.SUPPRESS)       # internal: the in_library_shards leg (a comma-separated device list)
    args = ap.parse_args()

    if args.child_in_library:
        return in_library_child([int(x) for x in args.child_in_library.split(",")], args.c5_batch, args.steps)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _relaunch(args.gpus)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import numpy as np
    import torch
    import torch.distributed as dist
    from libzkp_amd import _native, workloads as wl

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()) if world == 1 else "29500")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP prover has no CPU fallback")
    if os.environ.get("ZKP_BENCH_DEVICE") is not None:          # rehearsal only: several ranks on one GPU
        local_rank = int(os.environ["ZKP_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    on_gpu = args.dist_backend == "nccl"
    if use_dist:
        if on_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    gdev = dev if on_gpu else torch.device("cpu")

    L = _native.lib()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _native.check(L.zkp_hip_init(local_rank), "zkp_hip_init")
    keys = []
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):      # ONE trusted setup for every rank: the committed test keys
        blob = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
        keys.append((kind, blob))
        _native.check(L.zkp_hip_groth16_load_key(kind, blob, len(blob)), "zkp_hip_groth16_load_key")

    key_info = {}
    for kind, nm in ((0, "equality"), (1, "membership")):
        wb, un, tb = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
        _native.check(L.zkp_hip_groth16_key_info(kind, ctypes.byref(wb), ctypes.byref(un), ctypes.byref(tb)), "zkp_hip_groth16_key_info")
        key_info[nm] = {"radix": "2^%d%s" % (wb.value, " uneven (18 windows)" if un.value else ""), "table_bytes": tb.value}

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    class Gatherer:
        """The node-level process_batch hands every proof back: all_gather of the ranks' packed proofs (device-resident results of the
        staged batch `h`).  overlap = True: two buffer pairs, the gather of step k runs on RCCL's stream while step k + 1 is proved."""
        def __init__(self, h, overlap):
            cap = int(L.zkp_hip_batch_max_bytes(h))
            caps = [torch.zeros(1, dtype=torch.int64, device=gdev) for _ in range(world)]
            dist.all_gather(caps, torch.tensor([cap], dtype=torch.int64, device=gdev))
            slot = max(int(c.item()) for c in caps)
            nbuf = 2 if overlap else 1
            self.h, self.overlap, self.n = h, overlap, 0
            self.mine = [torch.zeros(slot, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
            self.all = [torch.zeros(world * slot, dtype=torch.uint8, device=gdev) for _ in range(nbuf)]
            self.pending = [None] * nbuf

        def gather(self):
            b = self.n % len(self.mine); self.n += 1
            if self.pending[b] is not None:
                self.pending[b].wait()                                # this buffer pair was handed to the gather two steps ago
            s = torch.cuda.current_stream()
            _native.check(L.zkp_hip_batch_device_results(self.h, 0, self.mine[b].data_ptr(), self.mine[b].numel(), None, None, ctypes.c_void_p(s.cuda_stream)), "device_results")
            if self.all[b].is_cuda:
                w = dist.all_gather_into_tensor(self.all[b], self.mine[b], async_op=True)
            else:
                s.synchronize()
                w = dist.all_gather_into_tensor(self.all[b], self.mine[b].cpu(), async_op=True)
            if self.overlap:
                self.pending[b] = w
            else:
                w.wait()                                              # (on the GPU this only orders torch's stream behind the collective ...)
                if self.all[b].is_cuda:
                    torch.cuda.current_stream().synchronize()         # ... so block the host as well: nothing of the next step may run under this gather

        def drain(self):
            for b in range(len(self.pending)):                       # every gather issued so far has landed before the clock is read
                if self.pending[b] is not None:
                    self.pending[b].wait(); self.pending[b] = None

    # ------------------------------------------------------------------ the contract's line: one 4096-op batch per rank
    n = args.batch
    ops, lists, seeds = wl.mixed_ops(n, 5 + rank)
    counts = {k: int((ops["kind"] == c).sum()) for k, c in (("range", 1), ("equality", 2), ("membership", 4), ("improvement", 5))}
    h = ctypes.c_void_p()
    _native.check(L.zkp_hip_batch_stage(n, P(ops), P(lists), P(seeds), ctypes.byref(h)), "zkp_hip_batch_stage")      # inputs resident in HBM from here on
    cap = int(L.zkp_hip_batch_max_bytes(h))
    G = Gatherer(h, overlap=True) if use_dist else None

    def step():
        _native.check(L.zkp_hip_batch_prove(h), "zkp_hip_batch_prove")
        if G is not None:
            G.gather()

    def timed_steps(k):
        ms = []
        t0 = time.perf_counter()
        for _ in range(k):
            ts = time.perf_counter()
            step()
            ms.append((time.perf_counter() - ts) * 1e3)
        if G is not None:
            G.drain()
        barrier()
        return time.perf_counter() - t0, ms

    for _ in range(args.warmup):
        step()
    if G is not None:
        G.drain()
    barrier()
    dt, step_ms = timed_steps(args.steps)                            # the timed region: no instrumentation inside the library

    # second pass, same K steps, with an event pair around every MSM launch: the launch durations behind `roofline`
    L.zkp_hip_profile_enable(1)
    for k in range(3):
        L.zkp_hip_profile_read_kernel(k, None, None, None, 1)
    dt_prof, _ = timed_steps(args.steps)
    prof = []
    for k in range(3):
        ms, launches, adds = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
        L.zkp_hip_profile_read_kernel(k, ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(adds), 1)
        prof.append((ms.value, launches.value, adds.value))
    L.zkp_hip_profile_enable(0)

    # correctness guard inside the bench: every op of the last step succeeded and the sizes are the expected ones
    out = np.zeros(cap, dtype=np.uint8); off = np.zeros(n + 1, dtype=np.uint64); st = np.zeros(n, dtype=np.int32)
    rc = _native.check(L.zkp_hip_batch_fetch(h, P(out), cap, P(off), P(st)), "zkp_hip_batch_fetch")
    assert rc == 0 and not st.any()
    lens = np.diff(off.astype(np.int64))
    assert (lens[ops["kind"] == 1] == 1478).all() and (lens[ops["kind"] == 2] == 298).all() and (lens[ops["kind"] == 4] == 430).all() and (lens[ops["kind"] == 5] > 2000).all()
    out_bytes = int(off[n])

    extra = {}
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    legs = not args.no_extra_legs and not under_profiler               # a profile of this command holds the contract's steps only

    def timed(f, reps):
        f()
        ts = []
        for _ in range(reps):
            t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
        return statistics.median(ts)

    if world == 1 and legs and not args.force_dist:
        try:                                                         # nothing in the legs beside the contract's line may cost that line
            hb_off = np.zeros(n + 1, dtype=np.uint64); hb_st = np.zeros(n, dtype=np.int32)
            t_hb = timed(lambda: _native.check(L.zkp_hip_process_batch(n, P(ops), P(lists), P(seeds), P(out), cap, P(hb_off), P(hb_st)), "process_batch"), 8)
            # two batches in flight (zkp_hip_batch_prove_async on two staged copies of the batch, alternately): how a server that feeds
            # batch after batch calls the library; reported beside the contract's line, never as `value`
            h_b = ctypes.c_void_p()
            _native.check(L.zkp_hip_batch_stage(n, P(ops), P(lists), P(seeds), ctypes.byref(h_b)), "stage")
            pair = [h, h_b]
            def pipelined(k=args.steps):
                _native.check(L.zkp_hip_batch_prove_async(pair[0]), "prove_async")
                for i in range(1, k):
                    _native.check(L.zkp_hip_batch_prove_async(pair[i & 1]), "prove_async")
                    _native.check(L.zkp_hip_batch_wait(pair[(i - 1) & 1]), "wait")
                _native.check(L.zkp_hip_batch_wait(pair[(k - 1) & 1]), "wait")
            pipelined(4)
            t_p0 = time.perf_counter(); pipelined(); t_p = time.perf_counter() - t_p0
            L.zkp_hip_batch_free(h_b)
            extra["two_batches_in_flight"] = {"value": args.steps * n / t_p, "unit": "proofs/s", "ms_per_step": t_p / args.steps * 1e3,
                                              "note": "the same K steps launched with zkp_hip_batch_prove_async on two staged batches alternately; not the contract's value"}
            extra["host_buffers"] = {"value": n / t_hb, "unit": "proofs/s", "ms_per_batch": t_hb * 1e3,
                                     "note": "the same batch through zkp_hip_process_batch: bucketing + validation + pinned staging + H2D + proving + D2H of %d proof bytes; not `value`" % out_bytes}
            other = {}
            for name, gen, cnt in (("C2_range_4096", wl.range_ops, 4096), ("C3_equality_4096", wl.equality_ops, 4096), ("C4_improvement_1024", wl.improvement_ops, 1024)):
                o2, l2, s2 = gen(cnt)
                h2 = ctypes.c_void_p()
                _native.check(L.zkp_hip_batch_stage(cnt, P(o2), P(l2), P(s2), ctypes.byref(h2)), "stage")
                t = timed(lambda: _native.check(L.zkp_hip_batch_prove(h2), "prove"), 5)
                L.zkp_hip_batch_free(h2)
                other[name] = {"proofs_per_s": cnt / t, "ms_per_batch": t * 1e3}
            extra["other_configs_staged"] = other
        except Exception as e:  # noqa: BLE001
            extra["extra_legs_error"] = repr(e)[:400]
        try:
            # One 16 384-op batch over N GPUs gives every GPU 16 384 / N ops of the same mix (batch.rs:123-131 fanned out; zkp_hip_plan_shards):
            # the step time of the mixed batch against its size on THIS GPU says what N GPUs can reach -- a batch has a fixed cost (the ~50
            # dependent launches of the Bulletproofs chain, the Groth16 latency kernels) that does not shrink with its share.
            sweep = {}
            for nb in (512, 1024, 2048, 4096, 8192, 16384):
                o_s, l_s, s_s = wl.mixed_ops(nb, 5)
                h_s = ctypes.c_void_p()
                _native.check(L.zkp_hip_batch_stage(nb, P(o_s), P(l_s), P(s_s), ctypes.byref(h_s)), "stage (sweep)")
                t_s = timed(lambda: _native.check(L.zkp_hip_batch_prove(h_s), "prove (sweep)"), 7 if nb <= 4096 else 4)
                cap_s = int(L.zkp_hip_batch_max_bytes(h_s))
                # the local half of the hand-over a sharded step pays: packed proofs of the staged batch -> the caller's device buffer
                buf_s = torch.zeros(cap_s, dtype=torch.uint8, device=dev)
                t_d = timed(lambda: _native.check(L.zkp_hip_batch_device_results(h_s, 0, buf_s.data_ptr(), cap_s, None, None, None), "device_results (sweep)"), 5)
                L.zkp_hip_batch_free(h_s)
                sweep[str(nb)] = {"ms_per_batch": t_s * 1e3, "proofs_per_s": nb / t_s, "packed_bytes_capacity": cap_s, "device_results_ms": t_d * 1e3}
            extra["batch_size_sweep"] = sweep
            t16 = sweep["16384"]["ms_per_batch"]
            pred = {}
            for N in (2, 4, 8):
                sh = sweep[str(16384 // N)]
                # blocking all_gather of every rank's packed proofs: each rank receives (N - 1) / N of the total through a ring bound by one
                # xGMI link; MODELLED (one GPU here), at 70 % of the link rate plus 20 us per ring step
                gather_ms = sweep["16384"]["packed_bytes_capacity"] * (N - 1) / N / (0.7 * XGMI_LINK_GBS * 1e9) * 1e3 + 0.02 * (N - 1)
                pred[str(N)] = {"share_ms": sh["ms_per_batch"], "device_results_ms": sh["device_results_ms"], "gather_ms_modelled": gather_ms,
                                "speedup": t16 / (sh["ms_per_batch"] + sh["device_results_ms"] + gather_ms)}
            extra["predicted_strong_scaling"] = {"t_16384_ms": t16, "by_n_gpus": pred,
                                                 "note": "t(16 384 ops on one GPU) / (t(16 384 / N ops on one GPU) + measured local hand-over + modelled ring all_gather); "
                                                         "a prediction from ONE GPU -- no multi-GPU run was made by this command"}
        except Exception as e:  # noqa: BLE001
            extra["sweep_leg_error"] = repr(e)[:400]
        try:                                                         # SURVEY 8f row N2 beside the contract's line: batched verification through the C ABI
            nv = 4096
            rngv = np.random.default_rng(5)
            vals = rngv.integers(0, 2**63, nv, dtype=np.uint64)
            eseed = np.frombuffer(rngv.bytes(32 * nv), dtype=np.uint8).copy()
            vbuf = np.zeros((nv, 298), dtype=np.uint8); vlen = np.zeros(nv, dtype=np.uint32); vst = np.zeros(nv, dtype=np.int32); vok = np.zeros(nv, dtype=np.uint8)
            _native.check(L.zkp_hip_prove_equality_batch(nv, P(vals), P(vals), P(eseed), P(vbuf), 298, P(vlen), P(vst)), "prove_equality")   # under the key loaded above
            t_ve = timed(lambda: _native.check(L.zkp_hip_verify_equality_batch(nv, P(vbuf), 298, P(vlen), P(vok)), "verify_equality"), 5)
            assert vok.all()
            rv = rngv.integers(0, 2**32, nv, dtype=np.uint64); rmn = np.zeros(nv, dtype=np.uint64); rmx = np.full(nv, 2**32, dtype=np.uint64)
            rseed = np.frombuffer(rngv.bytes(32 * nv), dtype=np.uint8).copy()
            rout = np.zeros((nv, 1478), dtype=np.uint8); rlen = np.zeros(nv, dtype=np.uint32); rst = np.zeros(nv, dtype=np.int32)
            _native.check(L.zkp_hip_prove_range_batch(nv, P(rv), P(rmn), P(rmx), 64, P(rseed), P(rout), 1478, P(rlen), P(rst)), "prove_range")
            t_vr = timed(lambda: _native.check(L.zkp_hip_verify_range_batch(nv, P(rout), 1478, P(rlen), P(rmn), P(rmx), P(vok)), "verify_range"), 5)
            assert vok.all()
            nbig = 65536                                             # above 8192 envelopes: one pairing check per call (g16_rlc.h)
            bbuf = np.ascontiguousarray(vbuf[np.arange(nbig) % nv]); blen = np.full(nbig, 298, dtype=np.uint32); bok = np.zeros(nbig, dtype=np.uint8)
            t_vb = timed(lambda: _native.check(L.zkp_hip_verify_equality_batch(nbig, P(bbuf), 298, P(blen), P(bok)), "verify_equality"), 3)
            assert bok.all()
            extra["verification_c_abi"] = {"equality_4096": {"envelopes_per_s": nv / t_ve, "ms_per_batch": t_ve * 1e3}, "range_4096": {"envelopes_per_s": nv / t_vr, "ms_per_batch": t_vr * 1e3},
                                           "equality_65536_one_pairing_check": {"envelopes_per_s": nbig / t_vb, "ms_per_batch": t_vb * 1e3},
                                           "note": "host buffers in, verdict bytes out; Groth16 on the Fq2 machine (DESIGN 6d), batches above 8192 envelopes through one "
                                                   "weighted pairing check (DESIGN R4.8); not part of the contract's value"}
        except Exception as e:  # noqa: BLE001
            extra["verification_leg_error"] = repr(e)[:400]
    L.zkp_hip_batch_free(h)

    # ------------------------------------------------------------------ BASELINE config 5: ONE 16 384-op batch sharded over the ranks
    if use_dist and legs:
        nt = args.c5_batch
        ops5, lists5, seeds5 = wl.mixed_ops(nt, 5)                        # the same batch on every rank
        owner = np.zeros(nt, dtype=np.uint32)
        _native.check(L.zkp_hip_plan_shards(nt, P(ops5), world, P(owner)), "zkp_hip_plan_shards")
        mine_ix = np.nonzero(owner == rank)[0]
        my_ops = ops5[mine_ix].copy()                                      # list_off still indexes the shared `lists5`
        my_seeds = np.ascontiguousarray(seeds5.reshape(nt, 32)[mine_ix]).ravel()
        h5 = ctypes.c_void_p()
        _native.check(L.zkp_hip_batch_stage(len(my_ops), P(my_ops), P(lists5), P(my_seeds), ctypes.byref(h5)), "stage (strong)")
        G5 = Gatherer(h5, overlap=False)

        def step5():
            _native.check(L.zkp_hip_batch_prove(h5), "zkp_hip_batch_prove")
            G5.gather()                                                    # blocking: every rank holds every proof when the step ends

        for _ in range(max(1, args.warmup)):
            step5()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step5()
        barrier()
        dt5 = time.perf_counter() - t0
        c5 = np.zeros(len(my_ops), dtype=np.int32); o5 = np.zeros(len(my_ops) + 1, dtype=np.uint64); b5 = np.zeros(int(L.zkp_hip_batch_max_bytes(h5)), dtype=np.uint8)
        assert _native.check(L.zkp_hip_batch_fetch(h5, P(b5), b5.size, P(o5), P(c5)), "fetch (strong)") == 0 and not c5.any()
        L.zkp_hip_batch_free(h5)
        t5 = torch.tensor([dt5], dtype=torch.float64, device=gdev)
        dist.all_reduce(t5, op=dist.ReduceOp.MAX)
        dt5 = float(t5.item())
        extra["strong_scaling_c5"] = {"value": nt * args.steps / dt5, "unit": "proofs/s", "ms_per_step": dt5 / args.steps * 1e3, "scaling": "strong", "n_gpus": world,
                                      "ops_in_the_one_batch": nt, "ops_on_rank_0": int(len(my_ops)),
                                      "note": "BASELINE config 5: one seeded %d-op mixed batch, per-variant contiguous slices from zkp_hip_plan_shards, each rank proves its slice, "
                                              "blocking all_gather of the packed proofs inside the clock (max over ranks); not the contract's value" % nt}

    # ------------------------------------------------------------------ the same contract steps with the opt-in key tables (child process)
    # `value` above is measured at the library's DEFAULT key tables (radix 2^13, ~28 GB for the two circuits).  A deployment that gives the
    # prover more HBM sets ZKP_HIP_G16_TABLE_BUDGET_MB: the same K steps under a 60 GB-per-key budget (radix 2^14 in its 18-window form, ~72 GB)
    # are reported beside the contract's line, never as `value`.
    if world == 1 and legs and not args.force_dist and not os.environ.get("ZKP_HIP_G16_TABLE_BUDGET_MB") and not os.environ.get("ZKP_HIP_G16_WBITS"):
        L.zkp_hip_shutdown()
        try:
            env = dict(os.environ, ZKP_HIP_G16_TABLE_BUDGET_MB="60000")
            pc = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline", "--no-extra-legs"],
                                capture_output=True, text=True, timeout=240, env=env)
            lines = [x for x in pc.stdout.splitlines() if x.startswith("{")]
            if pc.returncode == 0 and lines:
                cj = json.loads(lines[-1])
                extra["opt_in_key_tables"] = {"value": cj["value"], "unit": "proofs/s", "ms_per_step": cj["ms_per_step"], "g16_table_radix": cj.get("g16_table_radix"),
                                              "g16_table_bytes": cj.get("g16_table_bytes"), "environment": "ZKP_HIP_G16_TABLE_BUDGET_MB=60000",
                                              "note": "the contract's K steps again in a child process with the larger key tables opted in; not `value`"}
            else:
                extra["opt_in_key_tables"] = {"error": "child exited with %d: %s" % (pc.returncode, pc.stderr[-300:])}
        except Exception as e:  # noqa: BLE001
            extra["opt_in_key_tables"] = {"error": repr(e)[:300]}

    # ------------------------------------------------------------------ one process driving every visible GPU (the in-library multi-GPU path)
    # Runs in a CHILD process (started after this one has released the GPU: a new program, not an exec of this one): nothing that happens
    # in that leg -- it is the one path of this file that has never run on more than one physical GPU -- can cost the contract's line.
    ngpu = torch.cuda.device_count()
    if world == 1 and legs and not args.force_dist and (ngpu > 1 or os.environ.get("ZKP_BENCH_SHARDS")):
        devices = os.environ.get("ZKP_BENCH_SHARDS") or ",".join(str(i) for i in range(ngpu))
        L.zkp_hip_shutdown()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child-in-library", devices, "--c5-batch", str(args.c5_batch), "--steps", str(args.steps)],
                               capture_output=True, text=True, timeout=180)
            lines = [x for x in p.stdout.splitlines() if x.startswith("{")]
            extra["in_library_shards"] = json.loads(lines[-1]) if p.returncode == 0 and lines else {"error": "child exited with %d: %s" % (p.returncode, p.stderr[-400:])}
        except Exception as e:  # noqa: BLE001  (a timeout or anything else: reported, never fatal)
            extra["in_library_shards"] = {"error": repr(e)[:400]}

    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total = world * args.steps * n
        ed, g1, g2 = prof
        g1_avg_ms = g1[0] / max(1, g1[1])
        # algorithmic bytes one G1 MSM launch must move: an equality launch serves its 346 B/proof ops, a membership
        # launch its 598 B/proof ops; the G1 MSM of a circuit is two launches (A/B1 and the l/h sum), each credited half
        g1_launches_per_step = max(1.0, g1[1] / max(1, args.steps))
        algo_launch = (ALGO_BYTES["equality"] * counts["equality"] + ALGO_BYTES["membership16"] * counts["membership"]) / g1_launches_per_step
        achieved = algo_launch / (g1_avg_ms * 1e-3) / 1e9 if g1_avg_ms > 0 else 0.0
        g1_mad_rate = g1[2] * MADS_PER_G1_MADD / (g1[0] * 1e-3) / 1e12 if g1[0] > 0 else 0.0        # T mad/s
        traffic, traffic_src = None, None
        if os.path.exists(TRAFFIC_JSON):
            tj = json.load(open(TRAFFIC_JSON))
            traffic, traffic_src = tj.get("hbm_bytes_per_launch"), tj.get("source")
        alone = alone_durations()
        a_g1 = alone.get("k_msm_gather<G1Msm>")
        adds_per_launch = g1[2] / max(1, g1[1])
        alone_valu = None
        if a_g1:
            rate = adds_per_launch * MADS_PER_G1_MADD / (a_g1["avg_ms"] * 1e-3) / 1e12
            alone_valu = {"avg_launch_ms": a_g1["avg_ms"], "achieved": rate, "frac": rate / MAD_ISSUE_T,
                          "g_additions_per_s": adds_per_launch / (a_g1["avg_ms"] * 1e-3) / 1e9,
                          "fraction_of_bare_loop": adds_per_launch / (a_g1["avg_ms"] * 1e-3) / 1e9 / G1_BARE_LOOP_GADDS,
                          "hbm_frac": algo_launch / (a_g1["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "source": os.path.relpath(ALONE_CSV, ROOT),
                          "profiled_build": alone_build(), "this_build": csrc_sha16(), "stale": alone_build() != csrc_sha16()}
        floor = sum(v["avg_ms"] * v["launches_per_step"] for k, v in alone.items() if k in ("k_msm_gather<G1Msm>", "k_msm_gather<G2Msm>", ED_KERNEL, "k_g16_qap", "k_stark_prove"))
        res = {
            "metric": "proofs/sec (whole node) + ms/proof p50, 4096-proof mixed batch (range / equality / membership / improvement) per MI355X",
            "value": total / dt, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32/u64 integer limbs (10x25.5-bit GF(2^255-19), BN254 Fq on 9x29-bit limbs in the MSM loops and 10x26-bit elsewhere, 8x32-bit Fr and scalars mod l, 2x64-bit f128)", "data": "synthetic",
            "config": {"workload": "process_batch of %d mixed ops, i mod 4 = prove_range(v,0,2^32) / prove_equality / prove_membership(16-element set) / "
                                   "prove_improvement, seed 5 (BASELINE.md C5's mix at the metric's 4096-op size)" % n,
                       "ops_per_gpu_per_step": n, "ops_by_variant": counts, "proof_bytes_per_step": out_bytes,
                       "timed_region": "zkp_hip_batch_prove on a batch staged in HBM; packed proofs + offsets left in HBM; no profiling inside the library"
                                       + ("; RCCL all_gather of every rank's packed proofs, overlapped with the next step's proving, all landed before the clock stops" if use_dist else ""),
                       "sharding": "independent ops, one 4096-op batch per rank, no exchange inside the proving"},
            "ms_per_proof_p50": statistics.median(step_ms) / n,
            "ms_per_batch_p50": statistics.median(step_ms),
            "profiled_pass": {"ms_per_step": dt_prof / args.steps * 1e3, "note": "the same K steps again with an event pair around every MSM launch (zkp_hip_profile_enable): "
                                                                              "the source of the launch durations below; its step time shows what the instrumentation costs"},
            "kernel_floor_ms_per_step": floor if floor else None,
            "kernel_floor_source": {"file": os.path.relpath(ALONE_CSV, ROOT), "profiled_build": alone_build(), "this_build": csrc_sha16(), "stale": alone_build() != csrc_sha16(),
                                    "note": "sum of the GPU-filling kernels' durations from a serialised rocprofv3 counter pass of this command (tools/profile_round.sh), "
                                            "not measured by this run: `stale` says whether the kernel sources have changed since"},
            "g16_table_radix": {k: v["radix"] for k, v in key_info.items()}, "g16_table_bytes": sum(v["table_bytes"] for v in key_info.values()),
            "g16_table_policy": "library default (radix 2^13; ZKP_HIP_G16_TABLE_BUDGET_MB opts into larger tables)" if not (os.environ.get("ZKP_HIP_G16_TABLE_BUDGET_MB") or os.environ.get("ZKP_HIP_G16_WBITS"))
                                else "environment: ZKP_HIP_G16_TABLE_BUDGET_MB=%s ZKP_HIP_G16_WBITS=%s" % (os.environ.get("ZKP_HIP_G16_TABLE_BUDGET_MB"), os.environ.get("ZKP_HIP_G16_WBITS")),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "k_msm_gather<G1Msm>",
                         "avg_launch_ms": g1_avg_ms, "launches": g1[1], "algorithmic_bytes_per_launch": algo_launch,
                         "share_of_step": g1[0] / (dt_prof * 1e3) if world == 1 else None,
                         "note": "integer-ALU-bound kernel (SURVEY 8d): see roofline_valu; launch durations are taken on the kernel's own stream while the "
                                 "other variants' kernels share the GPU"},
            "roofline_valu": {"bound": "valu-int", "achieved": g1_mad_rate, "peak": MAD_ISSUE_T, "unit": "T v_mad_u64_u32/s", "frac": g1_mad_rate / MAD_ISSUE_T,
                              "kernel": "k_msm_gather<G1Msm>", "point_additions_per_step": g1[2] / max(1, args.steps), "mads_per_point_addition": MADS_PER_G1_MADD,
                              "g_additions_per_s": g1[2] / (g1[0] * 1e-3) / 1e9 if g1[0] > 0 else 0.0,
                              "fraction_of_bare_loop": (g1[2] / (g1[0] * 1e-3) / 1e9 / G1_BARE_LOOP_GADDS) if g1[0] > 0 else 0.0,
                              "alone": alone_valu,
                              "peak_source": "profiles/r02_fe_microbench.json: isolated v_mad_u64_u32 issue rate (the multiply-adds of the field products only; "
                                             "carries, masks and loads share the same issue slots: a pure product chain reaches 75 % of it, the bare addition loop "
                                             "(profiles/r04_g1_add_rate.jsonl) 75 %); launch durations are taken while the other variants' kernels share the GPU, "
                                             "`alone` from the serialised counter pass"},
            "other_msm_kernels": {
                "k_msm_gather<G2Msm>": {"avg_launch_ms": g2[0] / max(1, g2[1]), "launches": g2[1], "ms_per_step": g2[0] / max(1, args.steps),
                                        "point_additions_per_step": g2[2] / max(1, args.steps), "g_additions_per_s": g2[2] / (g2[0] * 1e-3) / 1e9 if g2[0] > 0 else None,
                                        "mads_per_point_addition": 4536},
                ED_KERNEL: {"avg_launch_ms": ed[0] / max(1, ed[1]), "launches": ed[1], "ms_per_step": ed[0] / max(1, args.steps),
                                     "point_additions_per_step": ed[2] / max(1, args.steps), "g_additions_per_s": ed[2] / (ed[0] * 1e-3) / 1e9 if ed[0] > 0 else None,
                                     "valu_frac": (ed[2] * MADS_PER_ED_MADD / (ed[0] * 1e-3) / 1e12 / MAD_ISSUE_T) if ed[0] > 0 else None}},
        }
        res.update(extra)
        if world == 1 and not args.no_cpu_baseline and not args.force_dist:      # the contract: rank 0 at N = 1 only
            threads = min(len(os.sched_getaffinity(0)), 32)
            try:
                res["cpu_baseline"] = cpu_baseline(args.cpu_sample, threads)
            except Exception as e:  # noqa: BLE001
                res["cpu_baseline"] = {"error": repr(e)[:400]}
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    L.zkp_hip_shutdown()


if __name__ == "__main__":
    main()
