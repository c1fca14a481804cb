#!/usr/bin/env python3
"""GPU probe: the warm-started search cut off after each of its phases (measurement builds
-DVISMA_COOP_STOP_AFTER=k, side libraries; results are garbage, the fold still runs): launch time as a function of
how far the queries get, at full load and without time stamps.   python tools/truncate_probe.py [ns nt]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %r)
from visma_amd import _lib, synth
ns, nt, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
c = _lib.Context(0)
c.set_clouds_f64(src, tgt)
c.set_nn_mode(_lib.NN_GRID)
c.iterate(T_gt, r, 4)
c.set_profiling(1)
c.iterate(T_gt, r, 3)
c.get_timing(reset=True)
c.iterate(T_gt, r, 30)
tm = c.get_timing(reset=True)
print(json.dumps({"stop_after_phase": k, "ns": ns, "nt": nt, "nn_us": tm["nn_ms"] / max(tm["nn_launches"], 1) * 1e3}), flush=True)
""" % ROOT


def main():
    from visma_amd import build
    a = [int(x) for x in sys.argv[1:] if x.isdigit()]
    ns, nt = (a + [262144, 4194304])[:2] if len(a) >= 2 else (262144, 4194304)
    for k in [int(x[2:]) for x in sys.argv if x.startswith("k=")] or (0, 1, 4, 5, 99):
        side = os.path.join(ROOT, "visma_amd", "lib", "libvisma_icp_stop%s.so" % str(k).replace("-", "m"))
        if not os.path.exists(side):
            build.build_lib(force=True, defines=("VISMA_COOP_STOP_AFTER=%d" % k,), out=side)
        if "--build-only" in sys.argv:
            continue
        env = dict(os.environ, VISMA_ICP_LIB=side)
        subprocess.run(["timeout", "120", sys.executable, "-c", CHILD, str(ns), str(nt), str(k)], env=env)


if __name__ == "__main__":
    main()
