#!/usr/bin/env python3
"""Time the mesh steps at the reference's evaluation size:
num_samples = min(500000, 100*|Fg|) (src/evaluation.cpp:326) against a scene of
`--copies` chairs.  Prints one JSON line.  `--cpu` also times igl::AABB (oracle/_ref)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visma_amd import _lib, synth  # noqa: E402


def scene(copies):
    m = np.load(os.path.join(ROOT, "tests", "golden", "mesh.npz"))
    V, F = m["V"], m["F"]
    rng = np.random.default_rng(3)
    Vs, Fs = [], []
    for i in range(copies):
        T = synth.make_T(synth.rot_y(rng.uniform(0, 6.28)), [2.0 * (i % 4), 0.0, 2.0 * (i // 4)])
        Vs.append(V @ T[:3, :3].T + T[:3, 3]); Fs.append(F + i * len(V))
    return np.concatenate(Vs), np.concatenate(Fs).astype(np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=10)
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--search", default="auto")
    a = ap.parse_args()
    V, F = scene(a.copies)
    n = a.samples or min(500000, 100 * len(F))
    T = synth.make_T(synth.rot_y(0.01), [0.003, 0.001, -0.002])
    Vt = V @ T[:3, :3].T + T[:3, 3]
    ctx = _lib.Context(0)
    ctx.sample_mesh(V, F, 1000, seed=1)                       # warm-up
    t0 = time.perf_counter(); pts = ctx.sample_mesh(V, F, n, quirks=False, seed=1); t_sample = time.perf_counter() - t0
    ctx.point_mesh_distance(pts[:1000], Vt, F)
    ctx.set_mesh_search("brute")
    ctx.point_mesh_distance(pts, Vt, F); brute_ms = ctx.last_mesh_kernel_ms()[0]
    ctx.set_mesh_search(a.search)
    t_dist = k_ms = b_ms = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); d2, face, cl = ctx.point_mesh_distance(pts, Vt, F)
        t_dist = min(t_dist, time.perf_counter() - t0)
        k, b = ctx.last_mesh_kernel_ms(); k_ms = min(k_ms, k); b_ms = min(b_ms, b)
    t_all = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); m = ctx.measure_surface_error(V, F, Vt, F, n, seed=1)
        t_all = min(t_all, time.perf_counter() - t0)
    out = dict(faces=len(F), samples=n, sample_s=t_sample, distance_s=t_dist, distance_kernel_ms=k_ms,
               build_ms=b_ms, brute_kernel_ms=brute_ms, brute_pairs_per_s=n * len(F) / (brute_ms * 1e-3),
               measure_surface_error_s=t_all, mean_error=m["mean"])
    if a.cpu:
        from oracle.oracle import Ref
        ref = Ref()
        t0 = time.perf_counter(); rd2, _, _ = ref.point_mesh_sqdist(pts, Vt, F); out["igl_aabb_s"] = time.perf_counter() - t0
        out["max_abs_d2_diff_vs_igl"] = float(np.abs(rd2 - d2).max())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
