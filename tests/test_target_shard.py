"""Target-sharded ranks (include/visma_icp.h: visma_icp_set_target_shard).

Two contexts on ONE GPU, each holding half of the target and all of the source, driven by
two host threads whose exchange callbacks meet at a barrier -- the same collectives per
iteration that RCCL performs between GPUs: with the exact grid search on every shard, MIN of the
f64 distance bits, MIN of the global index among the shards that hold that distance, SUM of the
38 statistics; with the fp32 brute-force kernel, MIN of the packed (fp32 d2, index) keys and the
SUM.  The result must be the single-context result: identical correspondences (global indices,
lowest index on ties), transform to summation order.
"""
import threading

import numpy as np
import pytest

from visma_amd import _lib, synth

# threads meeting at a barrier: never hang a test run
pytestmark = pytest.mark.timeout(300)


class Exchange:
    """In-process stand-in for the two all-reduces of a 2-rank communicator."""

    def __init__(self, n):
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slots = [None] * n
        self.calls = {"min": 0, "sum": 0}

    def _reduce(self, rank, a, op, kind):
        self.slots[rank] = a.copy()
        self.barrier.wait()
        out = self.slots[0].copy()
        for r in range(1, self.n):          # rank order: every rank gets bit-identical sums
            out = op(out, self.slots[r])
        self.barrier.wait()
        a[:] = out
        if rank == 0:
            self.calls[kind] += 1

    def minreduce(self, rank):
        return lambda a: self._reduce(rank, a, np.minimum, "min")

    def allreduce(self, rank):
        return lambda a: self._reduce(rank, a, np.add, "sum")


def run_sharded(src, tgt, radius, iters, nn_mode, cuts, normals=None, plane=False):
    n = len(cuts) - 1
    ex = Exchange(n)
    centre = tgt.mean(0)
    out = [None] * n
    err = []

    def worker(rank):
        try:
            ctx = _lib.Context(0)
            ctx.set_nn_mode(nn_mode)
            lo, hi = cuts[rank], cuts[rank + 1]
            ctx.set_target_shard(lo, len(tgt), centre)
            ctx.set_clouds_f64(src, tgt[lo:hi])
            if normals is not None:
                ctx.set_target_normals_f64(normals[lo:hi])
            ctx.set_minreduce(ex.minreduce(rank))
            ctx.set_allreduce(ex.allreduce(rank), rank, n)
            if plane:
                res = ctx.run_point_to_plane(None, radius, iters, 0, 0)
            else:
                res = ctx.run(None, radius, iters, 0, 0)
            out[rank] = (res, ctx.correspondence_index())
        except Exception as e:                  # pragma: no cover
            err.append(e)
            ex.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    [t.start() for t in th]
    [t.join() for t in th]
    if err:
        raise err[0]
    return out, ex


@pytest.mark.gpu
@pytest.mark.parametrize("nn_mode", [_lib.NN_BRUTE, _lib.NN_GRID])
def test_target_sharded_equals_single_context(lib, nn_mode):
    src, tgt, T_gt, radius = synth.make_pair(20000, 60000, motion="radius")
    tgt = np.concatenate([tgt, tgt[:500]])                 # exact duplicates across the shards: ties
    ref = _lib.Context(0)
    ref.set_nn_mode(nn_mode)
    if nn_mode == _lib.NN_BRUTE:
        # shards with a FORCED brute-force search exchange packed fp32 keys (the round-1 protocol):
        # compare them with the fp32 brute-force kernel, not with its exact flavour
        ref.set_search_precision("f32")
    ref.set_clouds_f64(src, tgt)
    want = ref.run(None, radius, 12, 0, 0)
    assert ref.search_mode_used() == ("exact" if nn_mode == _lib.NN_GRID else "f32")
    want_idx = ref.correspondence_index()
    cuts = [0, 23456, len(tgt)]                             # ragged shards
    out, ex = run_sharded(src, tgt, radius, 12, nn_mode, cuts)
    # per NN pass: one SUM, and one MIN (fp32 keys) or two (f64 distance, then index)
    assert ex.calls["sum"] == 13 and ex.calls["min"] == (26 if nn_mode == _lib.NN_GRID else 13)
    for res, idx in out:
        assert np.array_equal(idx, want_idx)                # global indices, lowest on ties
        assert res.num_correspondences == want.num_correspondences
        assert synth.rel_frobenius(res.transformation_, want.transformation_) < 1e-12
        assert abs(res.inlier_rmse_ - want.inlier_rmse_) < 1e-12
    assert np.array_equal(out[0][0].transformation_, out[1][0].transformation_)   # ranks agree to the bit


@pytest.mark.gpu
def test_target_sharded_three_ranks_point_to_plane_and_empty_shard(lib):
    src, tgt, T_gt, radius = synth.make_pair(6000, 30000, motion="radius")
    nrm = tgt / np.linalg.norm(tgt, axis=1, keepdims=True)
    ref = _lib.Context(0)
    ref.set_clouds_f64(src, tgt)
    ref.set_target_normals_f64(nrm)
    want = ref.run_point_to_plane(None, radius, 8, 0, 0)
    want_idx = ref.correspondence_index()
    cuts = [0, 11111, 11111, len(tgt)]                      # the middle rank owns nothing
    out, _ = run_sharded(src, tgt, radius, 8, _lib.NN_AUTO, cuts, normals=nrm, plane=True)
    for res, idx in out:
        assert np.array_equal(idx, want_idx)
        # Gauss-Newton with made-up normals is ill-conditioned: f64 summation order shows at 1e-10
        assert synth.rel_frobenius(res.transformation_, want.transformation_) < 1e-7


@pytest.mark.gpu
def test_target_shard_needs_an_exchange(lib):
    src, tgt, _, radius = synth.make_pair(2000, 8000, motion="radius")
    ctx = _lib.Context(0)
    ctx.set_target_shard(0, 16000, tgt.mean(0))
    ctx.set_clouds_f64(src, tgt)
    with pytest.raises(_lib.IcpError):
        ctx.run(None, radius, 2, 0, 0)                      # no RCCL communicator, no callbacks
    with pytest.raises(_lib.IcpError):
        ctx.set_target_shard(0, 2 ** 31 + 5, None)          # global indices must fit 31 bits


@pytest.mark.gpu
def test_source_sharded_equals_single_context(lib):
    """The default decomposition of bench.py --gpus N (each rank: a slice of the source, the whole
    target, ONE sum of the 38 statistics per pass), on the real HIP engine with threads as ranks."""
    src, tgt, T_gt, radius = synth.make_pair(30000, 100000, motion="radius")
    ref = _lib.Context(0)                     # default search precision on both sides (f64 here)
    ref.set_clouds_f64(src, tgt)
    want = ref.run(None, radius, 15, 0, 0)
    n = 3
    ex = Exchange(n)
    cuts = [0, 7001, 19000, len(src)]
    out, err = [None] * n, []

    def worker(rank):
        try:
            ctx = _lib.Context(0)
            ctx.set_clouds_f64(src[cuts[rank]:cuts[rank + 1]], tgt)
            ctx.set_global_source_count(len(src))
            ctx.set_allreduce(ex.allreduce(rank), rank, n)
            out[rank] = ctx.run(None, radius, 15, 0, 0)
        except Exception as e:                  # pragma: no cover
            err.append(e)
            ex.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not err, err
    assert ex.calls["sum"] == 16 and ex.calls["min"] == 0
    for res in out:
        assert res.num_correspondences == want.num_correspondences and res.fitness_ == want.fitness_
        assert synth.rel_frobenius(res.transformation_, want.transformation_) < 1e-12
        assert np.array_equal(res.transformation_, out[0].transformation_)      # ranks agree to the bit


@pytest.mark.gpu
def test_target_sharded_ranks_reproduce_the_reference_fuzz_cases(lib):
    """The exact search survives the sharding: the compiled reference's K and T on fuzz fixtures."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    from gen_fuzz import FIELDS, make_case
    G = np.load(os.path.join(here, "golden", "fuzz_ref.npz"))
    for i in range(0, len(G["ns"]), 9):
        src, tgt, init, r, iters = make_case({k: G[k][i] for k in FIELDS})
        n = len(tgt)
        cuts = [0, n // 3, n // 3 + n // 5, n]
        ex = Exchange(3)
        out = [None] * 3
        err = []

        def worker(rank):
            try:
                ctx = _lib.Context(0)
                ctx.set_target_shard(cuts[rank], n, tgt.mean(0))
                ctx.set_clouds_f64(src, tgt[cuts[rank]:cuts[rank + 1]])
                ctx.set_minreduce(ex.minreduce(rank))
                ctx.set_allreduce(ex.allreduce(rank), rank, 3)
                out[rank] = ctx.run(init, r, iters, 1e-6, 1e-6)
            except Exception as e:              # pragma: no cover
                err.append(e)
                ex.barrier.abort()
        th = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
        [t.start() for t in th]
        [t.join() for t in th]
        if err:
            raise err[0]
        for res in out:
            assert res.num_correspondences == int(G["ref_k"][i]), i
            assert synth.rel_frobenius(res.transformation_, G["ref_T"][i]) < 1e-9, i


@pytest.mark.gpu
@pytest.mark.parametrize("plane", [False, True])
def test_target_shard_device_loop_through_rccl(lib, plane):
    """A target-sharded rank in the device loop: keys, the two MIN all-reduces, the owners' moments, the sum
    and the solve all on the stream (RCCL), the host reads the state back every 8 passes.  One rank holding the
    whole target (a 1-rank communicator is what one GPU can run) must reproduce the unsharded run; the host
    loop through the same communicator too."""
    try:
        uid = _lib.comm_unique_id()
    except _lib.IcpError as e:                       # pragma: no cover
        pytest.skip("RCCL not loadable here: %s" % e)
    src, tgt, T_gt, radius = synth.make_pair(12000, 50000, motion="radius")
    tgt = np.concatenate([tgt, tgt[:300]])                 # exact ties
    nrm = tgt / np.linalg.norm(tgt, axis=1, keepdims=True)

    def run(ctx):
        return ctx.run_point_to_plane(None, radius, 10, 1e-9, 1e-9) if plane else ctx.run(None, radius, 10, 1e-9, 1e-9)

    ref = _lib.Context(0)
    ref.set_clouds_f64(src, tgt)
    if plane:
        ref.set_target_normals_f64(nrm)
    want = run(ref)
    want_idx = ref.correspondence_index()
    for device_loop in (False, True):
        ctx = _lib.Context(0)
        ctx.set_target_shard(0, len(tgt), tgt.mean(0))
        ctx.comm_init(0, 1, uid if not device_loop else _lib.comm_unique_id())
        ctx.set_device_loop(device_loop)
        ctx.set_clouds_f64(src, tgt)
        if plane:
            ctx.set_target_normals_f64(nrm)
        got = run(ctx)
        assert ctx.search_mode_used() == "exact"
        assert got.num_correspondences == want.num_correspondences and got.iterations == want.iterations
        assert synth.rel_frobenius(got.transformation_, want.transformation_) < (1e-7 if plane else 1e-12)
        assert np.array_equal(ctx.correspondence_index(), want_idx)
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["target", "source"])
def test_sharded_ranks_through_the_ring_search(lib, shard, monkeypatch):
    """Round 6: a radius that is large against the point spacing sends every rank's search to grid_ring.hip (forced here:
    the shards of a target each see their own occupancy).  Target shards exchange the f64 distances the ring kernel writes
    (d64_out) and the claimed indices; source shards only the sums.  The result must be the single context's on radius-sized
    cells: identical correspondences, transform to summation order."""
    src, tgt, T_gt, _ = synth.make_pair(6000, 50000, motion="fixed")
    radius = 0.15
    ref = _lib.Context(0)
    ref.set_nn_mode(_lib.NN_GRID)
    ref.set_ring_search(0)
    ref.set_clouds_f64(src, tgt)
    want = ref.run(None, radius, 10, 0, 0)
    want_idx = ref.correspondence_index()
    assert ref.search_kernel_used() != "ring"
    monkeypatch.setenv("VISMA_ICP_RING", "1")                # (read when the ranks' contexts are created)
    if shard == "target":
        out, ex = run_sharded(src, tgt, radius, 10, _lib.NN_GRID, [0, 21111, len(tgt)])
        assert ex.calls["sum"] == 11 and ex.calls["min"] == 22
        for res, idx in out:
            assert np.array_equal(idx, want_idx)
            assert res.num_correspondences == want.num_correspondences
            assert synth.rel_frobenius(res.transformation_, want.transformation_) < 1e-11
        assert np.array_equal(out[0][0].transformation_, out[1][0].transformation_)
        return
    n = 2
    ex = Exchange(n)
    cuts = [0, 2500, len(src)]
    out, kernels, err = [None] * n, [None] * n, []

    def worker(rank):
        try:
            ctx = _lib.Context(0)
            ctx.set_nn_mode(_lib.NN_GRID)
            ctx.set_clouds_f64(src[cuts[rank]:cuts[rank + 1]], tgt)
            ctx.set_global_source_count(len(src))
            ctx.set_allreduce(ex.allreduce(rank), rank, n)
            out[rank] = ctx.run(None, radius, 10, 0, 0)
            kernels[rank] = (ctx.search_kernel_used(), ctx.correspondence_index())
        except Exception as e:                  # pragma: no cover
            err.append(e)
            ex.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not err, err
    for r, res in enumerate(out):
        assert kernels[r][0] == "ring"
        assert np.array_equal(kernels[r][1], want_idx[cuts[r]:cuts[r + 1]])
        assert res.num_correspondences == want.num_correspondences
        assert synth.rel_frobenius(res.transformation_, want.transformation_) < 1e-11
    assert np.array_equal(out[0].transformation_, out[1].transformation_)
