"""bench.py with two ranks, one process each.
On two real GPUs: both shardings, through the peer-to-peer mailbox all-reduce and through RCCL (the MIN
all-reduce of the target-sharded mode included) -- skipped on a box with one GPU.
On ONE GPU (the box the suite normally runs on): the same command with both ranks on device 0 -- the whole
N > 1 path of the bench (gloo rendezvous, handle exchange, collective bring-up with fall-backs, barriers,
MAX-over-ranks timing, the weak-scaling line, scaling_workloads incl. the corpus counter) executes before the
driver's multi-GPU run does: source-sharded over the hipIpc mailboxes and over the host callback,
target-sharded over the host callbacks."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    from visma_amd import _lib
    return _lib.device_count()


def _bench(extra, env_extra, gpus):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    # the full result (every extra workload, the long phrases) goes to the side file the line names; the line on stdout
    # is the contract's summary: one line, under 4 KB
    side = os.path.join(tempfile.mkdtemp(prefix="visma_bench_"), "extras.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "6", "--warmup", "2",
           "--ns", "65536", "--nt", "262144", "--no-cpu-baseline", "--brute-steps", "0", "--f32-steps", "0",
           "--extras-file", side] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    assert len(lines[0]) < 4096, len(lines[0])
    line = json.loads(lines[0])
    full = json.load(open(side))
    assert line["extras"] == side and line["n_gpus"] == full["n_gpus"] == gpus
    assert abs(line["value"] - full["value"]) <= 1e-6 * full["value"] and "roofline" in line
    return full


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


def _same_registration(one, two):
    assert two["n_gpus"] == 2 and two["ranks_hold_identical_transforms"] is True
    assert abs(two["fitness"] - one["fitness"]) < 1e-12
    assert abs(two["err_vs_T_gt"] - one["err_vs_T_gt"]) < 1e-9
    assert abs(two["inlier_rmse"] - one["inlier_rmse"]) < 1e-12


@pytest.fixture(scope="module")
def one_rank(lib):
    return _bench(["--no-extras"], {}, 1)


@pytest.mark.gpu
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("comm,want", [("ipc", "hipipc"), ("torch", "callback")])
def test_two_source_sharded_ranks_on_one_gpu(lib, one_rank, comm, want):
    two = _bench(["--shard", "source"], {"VISMA_BENCH_COMM": comm, "VISMA_TEST_SHARE_GPU": "1"}, 2)
    _same_registration(one_rank, two)
    par = two["config"]["parallelism"].lower()
    assert want in par and "x2" in par and "gloo" in par, par
    assert two["weak_scaling"]["ns"] == 2 * 65536 and two["weak_scaling"]["fitness"] > 0.99
    sw = two["scaling_workloads"]
    assert set(sw) >= {"c4_grid_strong", "c4_weak", "c5_replicas"}
    assert sw["c5_replicas"]["icp_iterations_per_sec"] > 0 and "2 rank" in sw["c5_replicas"]["parallelism"]
    assert len(two["blocks"]["iterations_per_sec"]) == 7


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_two_source_sharded_ranks_keep_their_launches_alive(lib, one_rank):
    """What one rank per GPU does since round 4b: every rank's host loop keeps ONE launch of the certificate kernel alive,
    the folding workgroups of the ranks' launches exchange through the mailboxes pass after pass.  Ranks that share a GPU do
    not (their launches would wait for each other's compute units) unless told that they fit side by side: 2 x 128
    workgroups here."""
    two = _bench(["--shard", "source", "--no-weak", "--no-extras"],
                 {"VISMA_BENCH_COMM": "ipc", "VISMA_TEST_SHARE_GPU": "1", "VISMA_ICP_PERSIST_SHARED_GPU": "1",
                  "VISMA_ICP_PERSIST_RANKS": "1"}, 2)
    _same_registration(one_rank, two)
    par = two["config"]["parallelism"].lower()
    assert "hipipc" in par and "persistent launch" in par and "visma_icp_persist_ranks=1" in par, par
    launch = two["roofline"].get("launch", {})
    assert launch.get("persistent") is True and launch["passes_per_launch"] > 1, two["roofline"]


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_ranks_launch_once_per_pass_unless_asked(lib, one_rank):
    """The default for ranks since round 5 (first contact with real peers must meet the path that has run): one launch per
    pass, the exchange through the mailboxes inside it -- even where the launches would fit side by side; and the bench
    measures the opt-in mode as well (`ranks_persistent`), after `value`."""
    two = _bench(["--shard", "source", "--no-weak"],
                 {"VISMA_BENCH_COMM": "ipc", "VISMA_TEST_SHARE_GPU": "1", "VISMA_ICP_PERSIST_SHARED_GPU": "1"}, 2)
    _same_registration(one_rank, two)
    par = two["config"]["parallelism"].lower()
    assert "hipipc" in par and "one launch per pass" in par and "default for ranks" in par, par
    assert "launch" not in two["roofline"], two["roofline"]
    rp = two["ranks_persistent"]
    assert "error" not in rp and rp["persist_passes_timed"] > 0 and abs(rp["fitness"] - two["fitness"]) < 1e-12, rp


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_two_target_sharded_ranks_on_one_gpu(lib, one_rank):
    """north_star's decomposition: each rank holds half of the target; per pass a MIN exchange of the f64
    distances, a MIN exchange of the claimed indices, then the 38-double sum (host callbacks on one GPU)."""
    two = _bench(["--shard", "target", "--no-weak", "--no-extras"], {"VISMA_BENCH_COMM": "torch", "VISMA_TEST_SHARE_GPU": "1"}, 2)
    _same_registration(one_rank, two)
    assert "target-sharded x2" in two["config"]["parallelism"]


@pytest.mark.gpu
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("where", ["EXPORT", "INIT", "HANG"])
def test_one_ranks_failed_mailbox_moves_every_rank_to_the_next_transport(lib, one_rank, where):
    """The fall-back chain of bench.py: attach_comm.  Rank 1's hipIpc bring-up fails (injected: at
    the export, or after every rank has exported, when the peers' handles are mapped -- rank 0 HAS mapped by then and
    must let go again): both ranks must leave the mailboxes together and meet on the next transport that works here
    (RCCL refuses two ranks on one GPU: the host callback), and the registration must still be the one-rank one."""
    # (HANG, round 5: rank 1's mapping call never returns -- every bring-up step runs under a wall-clock limit, the rank
    #  counts it as failed after 5 s here, and the chain moves on as for a failure; the peer, whose own mapping waits for
    #  rank 1's handshake, is released by the same limit)
    env = ({"VISMA_BENCH_HANG_IPC_INIT_RANK": "1", "VISMA_BENCH_BRINGUP_TIMEOUT_S": "5"} if where == "HANG"
           else {"VISMA_BENCH_FAIL_IPC_%s_RANK" % where: "1"})
    two = _bench(["--shard", "source", "--no-weak", "--no-extras"], dict(env, VISMA_TEST_SHARE_GPU="1"), 2)
    _same_registration(one_rank, two)
    par = two["config"]["parallelism"].lower()
    assert "callback" in par and "hipipc" not in par and "x2" in par, par
