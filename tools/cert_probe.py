#!/usr/bin/env python3
"""GPU probe: the search launch, pass by pass, along one registration from the identity start -- kernel time (HIP
events), candidates examined and the share of queries the certificate of grid_coop.hip decided without a search --
with the certificate on and off (VISMA_ICP_CERT=0).   python tools/cert_probe.py [ns nt] [--overlap F] [--passes N]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visma_amd import _lib, synth  # noqa: E402


def arg(name, default, cast=float):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def main():
    a = [int(x) for x in sys.argv[1:] if x.isdigit()]
    ns, nt = a[:2] if len(a) >= 2 else (262144, 4194304)
    passes = arg("--passes", 24, int)
    overlap = arg("--overlap", 1.0)
    src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
    if overlap < 1.0:
        rng = np.random.default_rng(3)
        sel = rng.permutation(ns)[: int(ns * (1.0 - overlap))]
        src = src.copy()
        src[sel] += np.array([0.0, 4.0, 0.0])
    rows = {}
    for cert in (1, 0):
        os.environ["VISMA_ICP_CERT"] = str(cert)
        c = _lib.Context(0)
        os.environ.pop("VISMA_ICP_CERT")
        c.set_nn_mode(_lib.NN_GRID)
        c.set_clouds_f64(src, tgt)
        c.set_device_loop(False)
        c.set_profiling(1)
        T = np.eye(4)
        c.iterate(T, r, 1)                                     # grid build, buffers
        c.forget_winners()
        c.get_timing(reset=True)
        out = []
        for p in range(passes):
            T, res = c.iterate(T, r, 1)                        # one solve + the NN passes around it
            tm = c.get_timing(reset=True)
            nl = max(tm["nn_launches"], 1)
            out.append({"pass": p, "launches": tm["nn_launches"], "nn_us": 1e3 * tm["nn_ms"] / nl,
                        "cand_per_q": tm["grid_candidates"] / nl / ns, "certified": tm["grid_certified"] / nl / ns,
                        "kernel": c.search_kernel_used(), "K": res.num_correspondences})
        rows["cert" if cert else "nocert"] = out
        c.close()
    print("ns=%d nt=%d overlap=%.2f radius=%.5f" % (ns, nt, overlap, r))
    print(" pass |  cert: us  cand/q  certified |  no cert: us  cand/q | K")
    for x, y in zip(rows["cert"], rows["nocert"]):
        print(" %4d | %8.1f %7.2f %9.4f | %11.1f %7.2f | %d %s" % (x["pass"], x["nn_us"], x["cand_per_q"], x["certified"],
                                                                   y["nn_us"], y["cand_per_q"], x["K"], x["kernel"]))
    if "--json" in sys.argv:
        print(json.dumps(rows))


if __name__ == "__main__":
    main()
