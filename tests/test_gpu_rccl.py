"""The RCCL branch of bench.py at world size 1 (the driver's N > 1 runs need a multi-GPU node; this keeps the code path -- process-group
init on the GPU, zkp_hip_batch_device_results onto torch's stream, all_gather_into_tensor overlapped and blocking, the strong-scaling
leg's zkp_hip_plan_shards slicing -- executed by every GPU test run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gather_path_over_rccl_at_world_size_one():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1", "--batch", "512", "--c5-batch", "1024",
           "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [x for x in p.stdout.splitlines() if x.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["scaling"] == "weak"
    assert "all_gather" in r["config"]["timed_region"]
    s = r["strong_scaling_c5"]
    assert s["scaling"] == "strong" and s["value"] > 0 and s["ops_in_the_one_batch"] == 1024 and s["ops_on_rank_0"] == 1024
