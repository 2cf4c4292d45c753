"""N > 1 path on CPU: world_size-2 gloo run of the sharded batch driver with the oracle standing in as the
prover (checks partitioning, order preservation and the gather; the GPU prover itself is covered by -m gpu)."""
import os
import subprocess
import sys
import textwrap

from libzkp_amd.sharding import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import ctypes, os, sys
    import numpy as np
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import torch.distributed as dist
    from libzkp_amd.sharding import process_range_batch_sharded
    from util import oracle_prove, workload
    import __graft_entry__ as ge
    orc = ctypes.CDLL(ge.ORACLE_LIB); orc.zkp_oracle_init()
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    n = 7
    v, mn, mx, seeds = workload(n, 21)
    def prover(vals, mins, maxs, sd):
        a = lambda x: np.array(list(x), dtype=np.uint64)
        rc, out, lens, st = oracle_prove(orc, a(vals), a(mins), a(maxs), np.frombuffer(sd, dtype=np.uint8).copy(), threads=2)
        assert rc == 0
        return [out[i].tobytes() for i in range(len(vals))]
    got = process_range_batch_sharded(list(map(int, v)), list(map(int, mn)), list(map(int, mx)), seeds.tobytes(), prover=prover)
    rc, ref, lens, st = oracle_prove(orc, v, mn, mx, seeds, threads=2)
    assert len(got) == n and all(got[i] == ref[i].tobytes() for i in range(n)), "sharded result differs from the unsharded one"
    dist.barrier(); dist.destroy_process_group()
    sys.stdout.write("rank " + os.environ["RANK"] + " ok" + chr(10)); sys.stdout.flush()      # one write per rank: the two ranks share the pipe
""")


def test_world_size_two_gloo(tmp_path, oracle_c):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    import socket
    with socket.socket() as sk:                      # a free rendezvous port (a fixed one can still be in TIME_WAIT from the previous run)
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


MIXED_WORKER = textwrap.dedent("""
    import ctypes, os, sys
    import numpy as np
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import torch.distributed as dist
    from libzkp_amd.sharding import process_ops_sharded
    from libzkp_amd import api
    from util import oracle_prove
    from oracle.py import stark
    import __graft_entry__ as ge
    orc = ctypes.CDLL(ge.ORACLE_LIB); orc.zkp_oracle_init()
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    calls = []
    def prover(kind, sel, sd):                      # the oracle stands in for the GPU prover; same (kind, ops, seeds) contract
        calls.append((kind, len(sel)))
        if kind == "range":
            a = lambda k: np.array([o[k] for o in sel], dtype=np.uint64)
            rc, out, lens, st = oracle_prove(orc, a(1), a(2), a(3), np.frombuffer(sd, dtype=np.uint8).copy(), threads=2)
            assert rc == 0
            return [out[i].tobytes() for i in range(len(sel))]
        if kind == "improvement":
            return [stark.prove_improvement(o[1], o[2]) for o in sel]
        raise AssertionError(kind)
    ops = []
    for i in range(9):                               # interleaved variants, odd counts (uneven shards)
        ops.append(("range", 1000 + i, 0, 2**32))
        if i %% 2 == 0:
            ops.append(("improvement", 10 * i, 10 * i + 7 + i))
    seeds = bytes((7 * k + 3) %% 256 for k in range(32 * len(ops)))
    # one trusted setup per job: rank 0's key bytes reach every other rank
    from libzkp_amd.sharding import share_snark_keys
    installed = {}
    share_snark_keys([0, 1], export=lambda k: b"key-of-rank-0-circuit-%%d" %% k * (1000 + k), install=lambda k, b: installed.__setitem__(k, b))
    if int(os.environ["RANK"]) == 0:
        assert installed == {}
    else:
        assert installed == {0: b"key-of-rank-0-circuit-0" * 1000, 1: b"key-of-rank-0-circuit-1" * 1001}
    got = process_ops_sharded(ops, seeds, prover=prover)
    ref = api.prove_ops(ops, seeds, prover=prover)   # unsharded, same stand-in
    assert len(got) == len(ops) and all(g == r and g is not None for g, r in zip(got, ref)), "sharded mixed batch differs"
    mine = dict()
    for k, n in calls[:2]:
        mine[k] = n
    rank = int(os.environ["RANK"])
    assert mine == ({"range": 5, "improvement": 3} if rank == 0 else {"range": 4, "improvement": 2}), mine
    dist.barrier(); dist.destroy_process_group()
    sys.stdout.write("rank " + str(rank) + " mixed ok" + chr(10)); sys.stdout.flush()
""")


def test_world_size_two_gloo_mixed_batch(tmp_path, oracle_c):
    import socket
    script = tmp_path / "worker_mixed.py"
    script.write_text(MIXED_WORKER % {"root": ROOT})
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank 0 mixed ok" in out.stdout and "rank 1 mixed ok" in out.stdout
