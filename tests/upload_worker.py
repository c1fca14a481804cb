"""One registration printed as JSON (tests/test_upload_paths.py runs it with and without the raw upload)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visma_amd import _lib, synth  # noqa: E402

ns, nt, stride = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
src, tgt, T_gt, r = synth.make_pair(ns, nt, seed_t=91, seed_s=92, offset=[2.0, -1.0, 0.5], motion="radius")
kind = sys.argv[5] if len(sys.argv) > 5 else "f32exact"
# make_pair's coordinates are fp32-representable doubles (the upload then sends them as fp32 and widens them back
# on the device); "f64": values fp32 cannot hold everywhere; "mixed": only in the last 40 % of the target, so the
# upload switches from fp32 pieces to f64 pieces on the way
if kind != "f32exact":
    rng = np.random.default_rng(5)
    lo = 0 if kind == "f64" else int(nt * 0.6)
    tgt = tgt.copy()
    tgt[lo:] += rng.uniform(-1, 1, size=(nt - lo, 3)) * 1e-9
    if kind == "f64":
        src = src + rng.uniform(-1, 1, size=src.shape) * 1e-9
if stride > 3:                                    # strided views: the C ABI takes any stride >= 3
    S = np.zeros((ns, stride)); S[:, :3] = src
    T = np.zeros((nt, stride)); T[:, :3] = tgt
else:
    S, T = src, tgt
ctx = _lib.Context(0)
ctx.set_search_precision(sys.argv[4])
L = ctx.L
import ctypes as C
dp = C.POINTER(C.c_double)
rc = L.visma_icp_set_clouds_f64(ctx._h, S.ctypes.data_as(dp), ns, stride, T.ctypes.data_as(dp), nt, stride)
assert rc == 0, rc
ctx.ns = ns
res = ctx.run(None, r, 12, 0.0, 0.0)
idx = ctx.correspondence_index()
print(json.dumps({"T": np.asarray(res.transformation_).ravel().tolist(), "k": int(res.num_correspondences),
                  "rmse": float(res.inlier_rmse_), "idx_sum": int(idx.astype(np.int64).sum()),
                  "idx_hash": int((idx.astype(np.int64) * (np.arange(len(idx)) % 1009 + 1)).sum()),
                  "mode": ctx.search_mode_used()}))
