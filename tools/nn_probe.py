"""Time the NN-correspondence and reduction kernels at several sizes (GPU box)."""
import sys
import os
import time
import json

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visma_amd import _lib, synth


def main():
    sizes = [(5000, 20000), (16384, 65536), (65536, 262144), (65536, 1048576),
             (65536, 4194304), (262144, 4194304)]
    if len(sys.argv) > 1:
        sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
    ctx = _lib.Context(0)
    ctx.set_profiling(True)
    out = []
    for ns, nt in sizes:
        src, tgt, T, r = synth.make_pair(ns, nt, motion="radius")
        ctx.set_clouds_f64(src, tgt)
        for mode, name in ((_lib.NN_BRUTE, "brute"), (_lib.NN_GRID, "grid")):
            ctx.set_nn_mode(mode)
            ctx.set_device_loop(True)
            ctx.run(None, r, 1, 0, 0)            # warm (+ grid build)
            tb = ctx.get_timing(reset=True)
            iters = 5 if (ns * nt > 1e11 and name == "brute") else 20
            t0 = time.time()
            res = ctx.run(None, r, iters, 0, 0)
            wall = time.time() - t0
            tm = ctx.get_timing(reset=True)
            nn = tm["nn_ms"] / tm["nn_launches"]
            rd = tm["reduce_ms"] / tm["reduce_launches"]
            pairs = ns * nt
            ctx.set_device_loop(False)
            ctx.set_profiling(False)                 # wall time without the event records
            t0 = time.time()
            ctx.run(None, r, iters, 0, 0)
            wall_host = time.time() - t0
            ctx.set_profiling(True)
            ctx.set_device_loop(None)
            ctx.get_timing(reset=True)
            rec = dict(mode=name, ns=ns, nt=nt, radius=r, nn_ms=nn, reduce_ms=rd,
                       iter_ms_wall_hostloop=wall_host / (iters + 1) * 1e3,
                       build_ms=tb["aux_ms"], gpairs_per_s=pairs / nn / 1e6,
                       iter_ms_wall=wall / (iters + 1) * 1e3, K=res.num_correspondences,
                       T_hash=float(res.transformation_.sum()),
                       err_vs_gt=synth.rel_frobenius(res.transformation_, T))
            print(json.dumps(rec), flush=True)
            out.append(rec)
    return out


if __name__ == "__main__":
    main()
