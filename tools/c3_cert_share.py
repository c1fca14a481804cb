import os, sys
sys.path.insert(0, "/root/repo")
os.environ["VISMA_ICP_COOP_KERNEL"] = "cert"
import numpy as np
import bench
from visma_amd import _lib
objs, probs = bench.c3_problems()
ctx = _lib.Context(0)
batch = ctx.make_batch([p[:4] for p in probs])
ctx.run_batch(batch, max_iter=30)
ctx.set_profiling(1); ctx.get_timing(reset=True)
res = ctx.run_batch(batch, max_iter=30)
tm = ctx.get_timing(reset=True)
q = sum(len(p[0]) for p in probs)
print("launches", tm["nn_launches"], "queries per launch", q, "certified share over all launches %.3f" % (tm["grid_certified"] / (tm["nn_launches"] * q)),
      "iterations", sum(r.iterations for r in res), "nn_ms per launch %.3f" % (tm["nn_ms"] / tm["nn_launches"]))
its = np.array([r.iterations for r in res]); print("iterations per problem: min %d median %d max %d" % (its.min(), np.median(its), its.max()))
