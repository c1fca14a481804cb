#!/usr/bin/env python3
"""A/B timing of library builds / env variants at a few sizes (one process per variant).
    python tools/ab_probe.py name=ENV1=v,ENV2=v ...       (name "base" = no env)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, time
import numpy as np
sys.path.insert(0, %r)
from visma_amd import _lib, synth
name = sys.argv[1]
sizes = [(5000, 20000, 60), (65536, 1048576, 40), (262144, 4194304, 40)]
prec = os.environ.get("PROBE_PREC", "exact")
for ns, nt, steps in sizes:
    src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
    c = _lib.Context(0)
    c.set_search_precision(prec)
    c.set_clouds_f64(src, tgt)
    c.set_nn_mode(_lib.NN_GRID)
    c.set_profiling(1)
    c.iterate(np.eye(4), r, 3)
    c.get_timing(reset=True)
    c.iterate(np.eye(4), r, steps)
    tm = c.get_timing(reset=True)
    c.set_profiling(0)
    c.iterate(np.eye(4), r, 3)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        T2, last = c.iterate(np.eye(4), r, steps)
        best = min(best, time.perf_counter() - t0)
    print(json.dumps({"cfg": name, "ns": ns, "us_per_iter": best / steps * 1e6,
                      "nn_us": tm["nn_ms"] / max(tm["nn_launches"], 1) * 1e3,
                      "fold_us": tm["reduce_ms"] / max(tm["reduce_launches"], 1) * 1e3,
                      "K": last.num_correspondences, "mode": c.search_mode_used(),
                      "Tsum": float(np.abs(T2).sum())}), flush=True)
    c.close()
''' % ROOT


def main():
    for spec in sys.argv[1:]:
        name, _, envs = spec.partition("=")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        out = subprocess.run([sys.executable, "-c", CHILD, name], env=env, capture_output=True, text=True)
        sys.stdout.write(out.stdout)
        if out.returncode != 0:
            sys.stdout.write("FAILED %s: %s\n" % (name, out.stderr[-600:]))
    sys.stdout.flush()


if __name__ == "__main__":
    main()
