"""The one-shot all-reduce through IPC-mapped mailboxes, with real processes (one HIP context each):
source-sharded ranks must hold the single-GPU result, and identical transforms on every rank.
On a box with fewer GPUs than ranks the ranks share device 0 (the mailbox mechanism is the same;
only the wire differs)."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _run(world, device_loop, persist_on_shared_gpu=False):
    import ipc_worker
    # (ranks that share a device never keep persistent launches alive side by side -- except when a test says its clouds
    #  are small enough for all of them to be resident together)
    # (round 5: ranks keep their launches alive only when asked to -- VISMA_ICP_PERSIST_RANKS=1 --, one launch per pass
    #  with the exchange inside it is their default)
    if persist_on_shared_gpu:
        os.environ["VISMA_ICP_PERSIST_SHARED_GPU"] = "1"
        os.environ["VISMA_ICP_PERSIST_RANKS"] = "1"
    else:
        os.environ.pop("VISMA_ICP_PERSIST_SHARED_GPU", None)
        os.environ.pop("VISMA_ICP_PERSIST_RANKS", None)
    from visma_amd import _lib, synth
    ndev = int(os.environ.get("VISMA_TEST_NDEV", "0")) or 1
    ndev = max(ndev, _lib.device_count())
    mpc = mp.get_context("spawn")
    pipes, procs = [], []
    for rank in range(world):
        a, b = mpc.Pipe()
        p = mpc.Process(target=ipc_worker.main, args=(rank, world, b, rank % ndev, device_loop))
        p.start()
        pipes.append(a)
        procs.append(p)
    try:
        handles = []
        for a in pipes:
            assert a.poll(120), "a rank did not export its mailbox"
            handles.append(a.recv())
        for a in pipes:
            a.send(handles)
        outs = []
        for a in pipes:
            assert a.poll(300), "a rank did not finish"
            outs.append(a.recv())
        for a in pipes:
            a.send(b"bye")
    finally:
        os.environ.pop("VISMA_ICP_PERSIST_SHARED_GPU", None)
        os.environ.pop("VISMA_ICP_PERSIST_RANKS", None)
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    src, tgt, T_gt, r = synth.make_pair(6000, 24000, motion="radius")
    one = _lib.Context(0)
    one.set_clouds_f64(src, tgt)
    w = one.run(None, r, 12, 0.0, 0.0)
    w.early = [one.run(None, r, 3 + 2 * k, 3e-2, 3e-2) for k in range(5)]
    return outs, w


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,device_loop,persist", [(2, False, False), (3, False, False), (2, True, False), (2, False, True), (3, False, True)])
def test_ranks_in_processes_hold_the_single_gpu_result(lib, world, device_loop, persist):
    """persist: the ranks' host loops each keep ONE launch alive across their passes (what one rank per GPU does; here
    the ranks share GPU 0 and say so is fine): the folding workgroup of every launch exchanges with the peers' launches
    through the mailboxes pass after pass -- same results, and the launches must really have run."""
    from visma_amd import synth
    outs, w = _run(world, device_loop, persist)
    for o in outs:
        launches, passes, aborts = o[6]
        assert aborts == 0
        assert (launches > 0 and passes > launches) if persist else launches == 0, o[6]
    outs = [o[:6] for o in outs]
    for T, k, fit, rmse, T2, early in outs:
        assert k == w.num_correspondences
        assert synth.rel_frobenius(T, w.transformation_) < 1e-12
        assert abs(fit - w.fitness_) < 1e-15
        assert any(it < 3 + 2 * j for j, (it, _, _) in enumerate(early))      # some did stop early
        for (it, kk, Te), ref in zip(early, w.early):
            assert it == ref.iterations and kk == ref.num_correspondences
            assert synth.rel_frobenius(Te, ref.transformation_) < 1e-12
    for T, k, fit, rmse, T2, early in outs[1:]:
        assert np.array_equal(T, outs[0][0])          # rank-ordered sums: bit-identical on every rank
        assert np.array_equal(T2, outs[0][4])
        for a, b in zip(early, outs[0][5]):
            assert np.array_equal(a[2], b[2])
