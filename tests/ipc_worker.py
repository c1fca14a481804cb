"""One rank of the peer-to-peer all-reduce test (tests/test_ipc_allreduce.py): a process with its own
HIP context, source-sharded, exchanging its mailbox handle with the others through the parent."""
import os
import sys

import numpy as np


def main(rank, world, conn, device, device_loop):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from visma_amd import _lib, synth
    src, tgt, T_gt, r = synth.make_pair(6000, 24000, motion="radius")
    lo, hi = (len(src) * rank) // world, (len(src) * (rank + 1)) // world
    ctx = _lib.Context(device)
    ctx.set_device_loop(device_loop)
    ctx.set_clouds_f64(src[lo:hi], tgt)
    ctx.set_global_source_count(len(src))
    conn.send(ctx.comm_ipc_export())
    handles = conn.recv()
    ctx.comm_ipc_init(rank, world, handles)
    ctx.set_profiling(1)                          # (counts the persistent launches, if any ran)
    ctx.get_timing(reset=True)
    res = ctx.run(None, r, 12, 0.0, 0.0)
    # registrations that STOP EARLY (loose stop test), again and again: in the device loop the launches
    # queued after convergence return without exchanging -- they must not use up exchange numbers, or the
    # next registration's first exchange lands in the mailbox half a slow peer is still reading
    early = []
    for k in range(5):
        e = ctx.run(None, r, 3 + 2 * k, 3e-2, 3e-2)
        early.append((int(e.iterations), int(e.num_correspondences), np.asarray(e.transformation_)))
    T2, last = ctx.iterate(np.eye(4), r, 5)
    tm = ctx.get_timing()
    conn.send((np.asarray(res.transformation_), int(res.num_correspondences), float(res.fitness_),
               float(res.inlier_rmse_), np.asarray(T2), early, (tm["persist_launches"], tm["persist_passes"], tm["persist_aborts"])))
    conn.recv()                                   # keep the mailbox alive until every rank is done
    ctx.close()
