"""Worker for test_distributed_gloo.py: one rank of a source-sharded ICP (CPU, gloo)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    from oracle.oracle import Oracle
    from oracle_engine import OracleEngine
    from visma_amd import synth

    out_path, mode = sys.argv[1], sys.argv[2]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    src, tgt, _, _ = synth.make_pair(3001, 7000)       # odd count: ragged shards
    lo, hi = (len(src) * rank) // world, (len(src) * (rank + 1)) // world
    eng = OracleEngine(Oracle())
    ctx = eng.context()
    ctx.set_clouds_f64(src[lo:hi], tgt)                # full target on every rank
    ctx.set_global_source_count(len(src))

    def allreduce(a):                                  # ONE all-reduce of 38 f64 per iteration
        t = torch.from_numpy(a)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    ctx.set_allreduce(allreduce, rank, world)
    if mode == "p2p":
        r = ctx.run(None, 0.075, 12, 0.0, 0.0)
    elif mode == "term":
        r = ctx.run(None, 0.075, 40, 1e-6, 1e-6)
    else:
        r = ctx.run(None, 0.075, 12, 0.0, 0.0, solver=1)
    # every rank must hold the SAME transform (each solves the same reduced system)
    Ts = [torch.zeros(16, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(Ts, torch.from_numpy(r.transformation_.ravel().copy()))
    same = all(torch.equal(Ts[0], t) for t in Ts)
    if rank == 0:
        np.savez(out_path, T=r.transformation_, k=r.num_correspondences, fitness=r.fitness_,
                 rmse=r.inlier_rmse_, iters=r.iterations, same=same, nn_calls=eng.calls["nn"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
