#!/usr/bin/env python3
"""Generate tests/golden/fuzz_ref.npz: random registrations run through the COMPILED REFERENCE
(oracle/_ref: Open3D's RegistrationICP + KDTreeFlann built from /root/reference by oracle/Makefile).

Only the recipe of each case is stored (the clouds are re-generated from the seeds by
visma_amd.synth, a counter-based Philox stream: identical on every machine) together with what the
reference returned: final transformation, correspondence count, fitness, inlier rmse.
Cases cover partial overlap (a slab of the source removed), scene offsets of 0 / 1 / 10 m, radii
of 0.7 .. 3 x the default, 1 .. 40 iterations with the reference's relative stop test.

    python tests/golden/gen_fuzz.py [N=64] [seed=20260929]
Runs in THIS container only (needs oracle/_ref); the .npz travels."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from visma_amd import synth  # noqa: E402
from oracle.oracle import Ref  # noqa: E402


def make_case(p):
    """Clouds + arguments of one case from its stored recipe (shared with the GPU test)."""
    src, tgt, T_gt, r = synth.make_pair(int(p["ns"]), int(p["nt"]), seed_t=int(p["seed_t"]), seed_s=int(p["seed_s"]),
                                        noise=float(p["noise"]), motion="radius")
    r *= float(p["rscale"])
    if p["cut"] > 0:                                   # partial overlap: drop the source points of a slab
        axis = int(p["cut_axis"])
        lo = np.quantile(src[:, axis], float(p["cut"]))
        src = src[src[:, axis] >= lo]
    off = np.asarray(p["offset"], dtype=np.float64)
    src = src + off
    tgt = tgt + off
    init = synth.make_T(synth.rot_y(float(p["yaw"])), np.asarray(p["shift"], dtype=np.float64))
    # the initial guess acts about the scene offset, like the clouds
    init[:3, 3] += off - init[:3, :3] @ off
    return src, tgt, init, r, int(p["iters"])


FIELDS = ("ns", "nt", "seed_t", "seed_s", "noise", "rscale", "cut", "cut_axis", "offset", "yaw", "shift", "iters")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260929)
    ref = Ref()
    rec = {k: [] for k in FIELDS}
    out = {k: [] for k in ("T", "k", "fitness", "rmse")}
    for it in range(n):
        p = {
            "ns": int(rng.integers(500, 8000)), "nt": int(rng.integers(2000, 40000)),
            "seed_t": int(rng.integers(1 << 30)), "seed_s": int(rng.integers(1 << 30)),
            "noise": 10.0 ** rng.uniform(-4, -2.5), "rscale": rng.uniform(0.7, 3.0),
            "cut": float(rng.choice([0.0, 0.0, 0.3, 0.6])), "cut_axis": int(rng.integers(3)),
            "offset": rng.standard_normal(3) * rng.choice([0.0, 1.0, 10.0]),
            "yaw": rng.uniform(-0.02, 0.02), "shift": rng.standard_normal(3) * 0.004,
            "iters": int(rng.integers(1, 40)),
        }
        src, tgt, init, r, iters = make_case(p)
        w = ref.registration_icp(src, tgt, r, init=init, max_iter=iters)
        for k in FIELDS:
            rec[k].append(p[k])
        out["T"].append(np.asarray(w.T)); out["k"].append(w.k); out["fitness"].append(w.fitness); out["rmse"].append(w.rmse)
        print("case %2d: ns=%d nt=%d r=%.4g iters=%d  K=%d fitness=%.4f" % (it, len(src), len(tgt), r, iters, w.k, w.fitness))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_ref.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in rec.items()}, **{"ref_" + k: np.asarray(v) for k, v in out.items()})
    print("wrote", path)


if __name__ == "__main__":
    main()
