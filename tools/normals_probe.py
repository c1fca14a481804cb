#!/usr/bin/env python3
"""GPU probe of visma_icp_estimate_normals against the compiled reference on the host cores.
    python tools/normals_probe.py [--cpu]   -> one JSON line per (size, search)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visma_amd import _lib, synth  # noqa: E402

ctx = _lib.Context(0)
ref = None
if "--cpu" in sys.argv:
    from oracle.oracle import Ref
    ref = Ref()
for n in (100000, 1 << 20, 1 << 22):
    pts = synth.surface_points(n, 9) + np.random.default_rng(3).normal(size=(n, 3)) * 1e-3
    r = synth.default_radius(n)
    for name, kw in (("knn30", dict(knn=30)), ("hybrid(2r,30)", dict(knn=30, radius=2 * r)), ("radius(r)", dict(knn=None, radius=r))):
        ctx.estimate_normals(pts[:2000], **kw)
        t = time.perf_counter(); got = ctx.estimate_normals(pts, **kw); gpu = time.perf_counter() - t
        row = {"n": n, "search": name, "radius": r, "gpu_ms_end_to_end": gpu * 1e3}
        if ref is not None and n <= (1 << 20):
            t = time.perf_counter(); want = ref.estimate_normals(pts, **kw); row["cpu_reference_ms"] = (time.perf_counter() - t) * 1e3
            err = np.abs(got - want).max(1)
            row["max_abs_diff"] = float(err.max()); row["q999_abs_diff"] = float(np.quantile(err, 0.999))
            row["cpu_threads"] = os.cpu_count()
        print(json.dumps(row), flush=True)
