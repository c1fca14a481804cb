"""CPU: the N > 1 path with world_size 2 over gloo.

Source-sharded ICP: each rank holds a slice of the source and the whole
target, reduces its slice to the 38 f64 normal-equation accumulators, and ONE
all-reduce per iteration sums them (RCCL on the GPU box, gloo here through the
visma_icp_set_allreduce hook).  The sharded run must give the single-process
answer, on every rank, bit-identical across ranks.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle_engine import OracleEngine
from visma_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(tmp_path, mode, world=2):
    out = str(tmp_path / ("dist_%s.npz" % mode))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "dist_worker.py"), out, mode]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return np.load(out)


@pytest.mark.parametrize("mode", ["p2p", "term", "gn"])
def test_source_sharded_equals_single_process(lib, oracle, tmp_path, mode):
    d = _launch(tmp_path, mode)
    src, tgt, _, _ = synth.make_pair(3001, 7000)
    ctx = OracleEngine(oracle).context()
    ctx.set_clouds_f64(src, tgt)
    if mode == "p2p":
        r = ctx.run(None, 0.075, 12, 0.0, 0.0)
    elif mode == "term":
        r = ctx.run(None, 0.075, 40, 1e-6, 1e-6)
    else:
        r = ctx.run(None, 0.075, 12, 0.0, 0.0, solver=1)
    assert bool(d["same"])                                   # identical on all ranks
    assert int(d["k"]) == r.num_correspondences             # global K, not the shard's
    assert abs(float(d["fitness"]) - r.fitness_) < 1e-12    # K / NS_total
    assert int(d["iters"]) == r.iterations
    assert int(d["nn_calls"]) == r.nn_passes
    # only the summation order of the 38 accumulators differs
    assert synth.rel_frobenius(d["T"], r.transformation_) < 1e-12
    assert abs(float(d["rmse"]) - r.inlier_rmse_) < 1e-12
