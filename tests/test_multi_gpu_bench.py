"""bench.py on two real GPUs, one process each: both shardings, through the peer-to-peer mailbox
all-reduce and through RCCL (the MIN all-reduce of the target-sharded mode included).  Skips on a
box with one GPU (the driver's multi-GPU run and the gloo dry runs of test_distributed_gloo.py cover
the same code there)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    from visma_amd import _lib
    return _lib.device_count()


def _bench(extra, env_extra, gpus):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "6", "--warmup", "2",
           "--ns", "65536", "--nt", "262144", "--no-cpu-baseline", "--brute-steps", "0", "--f32-steps", "0"] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("shard,comm", [("source", "ipc"), ("source", "rccl"), ("target", "rccl")])
def test_two_ranks_match_one(lib, shard, comm):
    if _ngpu() < 2:
        pytest.skip("needs two GPUs")
    one = _bench([], {}, 1)
    two = _bench(["--shard", shard, "--no-weak"], {"VISMA_BENCH_COMM": comm}, 2)
    assert two["n_gpus"] == 2 and two["ranks_hold_identical_transforms"] is True
    want = {"ipc": "hipipc", "rccl": "rccl"}[comm]
    assert want in two["config"]["parallelism"].lower(), two["config"]["parallelism"]
    # the same registration: same fitness, same distance to the ground-truth motion
    assert abs(two["fitness"] - one["fitness"]) < 1e-12
    assert abs(two["err_vs_T_gt"] - one["err_vs_T_gt"]) < 1e-9
    assert abs(two["inlier_rmse"] - one["inlier_rmse"]) < 1e-12
