"""Exploratory: every entry point of the Python wrapper with empty / one-point / degenerate inputs.  Prints what each call
did (a value, or the library's error); nothing here may crash or hang.  python tools/edge_probe.py"""
import sys
import traceback

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from visma_amd import _lib, synth  # noqa: E402

src, tgt, T_gt, r = synth.make_pair(3000, 9000, motion="fixed")
E = np.zeros((0, 3))
ONE = np.array([[0.1, 0.2, 0.3]])


def attempt(name, f):
    try:
        v = f()
        print("ok   ", name, "->", str(v)[:110].replace("\n", " "), flush=True)
    except _lib.IcpError as e:
        print("error", name, "->", str(e)[:140], flush=True)
    except Exception as e:          # noqa: BLE001
        print("PYERR", name, "->", type(e).__name__, str(e)[:140], flush=True)
        traceback.print_exc()


def ctx():
    c = _lib.Context(0)
    return c


for mode_name, mode in (("auto", _lib.NN_AUTO), ("grid", _lib.NN_GRID), ("brute", _lib.NN_BRUTE)):
    for sname, s, tname, t in (("empty", E, "full", tgt), ("full", src, "empty", E), ("empty", E, "empty", E), ("one", ONE, "one", ONE + 0.01),
                               ("one", ONE, "full", tgt), ("full", src, "one", ONE)):
        def run_all():
            c = ctx()
            c.set_nn_mode(mode)
            c.set_clouds_f64(s, t)
            out = []
            c.nn_pass(np.eye(4), 0.1)
            out.append(("K", c.reduce()[0]))
            out.append(("corr", len(c.get_correspondences()[0])))
            out.append(("run", c.run(None, 0.1, 3).num_correspondences))
            c.set_device_loop(True)
            out.append(("devloop", c.run(None, 0.1, 3).num_correspondences))
            c.set_device_loop(False)
            out.append(("sweep", c.run_yaw_sweep(4, 0.1, 3)[0].num_correspondences))
            T, res = c.iterate(None, 0.1, 2)
            out.append(("iterate", res.num_correspondences))
            c.close()
            return out
        attempt("%s: source %s, target %s" % (mode_name, sname, tname), run_all)

c = ctx()
attempt("voxel_down_sample empty", lambda: len(c.voxel_down_sample(E, 0.05)[0]))
attempt("voxel_down_sample one", lambda: len(c.voxel_down_sample(ONE, 0.05)[0]))
attempt("estimate_normals empty", lambda: c.estimate_normals(E, knn=10).shape)
attempt("estimate_normals one", lambda: c.estimate_normals(ONE, knn=10).shape)
attempt("estimate_normals two", lambda: c.estimate_normals(np.vstack([ONE, ONE + 0.1]), knn=10).shape)
V = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0]])
F = np.array([[0, 1, 2]], dtype=np.int32)
attempt("sample_mesh n=0", lambda: c.sample_mesh(V, F, 0).shape)
attempt("sample_mesh no faces", lambda: c.sample_mesh(V, np.zeros((0, 3), np.int32), 10).shape)
attempt("point_mesh_distance empty P", lambda: c.point_mesh_distance(E, V, F))
attempt("point_mesh_distance no faces", lambda: c.point_mesh_distance(ONE, V, np.zeros((0, 3), np.int32)))
attempt("batch with an empty problem", lambda: [x.num_correspondences for x in c.run_batch([(src, tgt, np.eye(4), 0.1), (E, tgt, np.eye(4), 0.1), (src, E, np.eye(4), 0.1)], 3)])
attempt("batch of nothing", lambda: c.run_batch([], 3))


def plane():
    d = ctx()
    d.set_clouds_f64(src, tgt)
    d.set_target_normals_f64(np.tile([0.0, 0, 1], (len(tgt), 1)))
    return d.run_point_to_plane(None, 0.1, 3).num_correspondences


attempt("point-to-plane", plane)


def plane_empty():
    d = ctx()
    d.set_clouds_f64(src, E)
    d.set_target_normals_f64(E)
    return d.run_point_to_plane(None, 0.1, 3).num_correspondences


attempt("point-to-plane, empty target", plane_empty)
attempt("radius 0", lambda: (c.set_clouds_f64(src, tgt), c.run(None, 0.0, 3).num_correspondences)[1])
attempt("radius nan", lambda: (c.set_clouds_f64(src, tgt), c.run(None, float("nan"), 3).num_correspondences)[1])
attempt("radius inf", lambda: (c.set_clouds_f64(src, tgt), c.run(None, float("inf"), 3).num_correspondences)[1])
attempt("radius 1e30", lambda: (c.set_clouds_f64(src, tgt), c.run(None, 1e30, 2).num_correspondences)[1])
attempt("nan in the source", lambda: (c.set_clouds_f64(np.vstack([src[:10], [[np.nan, 0, 0]]]), tgt), c.run(None, 0.1, 2).num_correspondences)[1])
attempt("inf in the target", lambda: (c.set_clouds_f64(src, np.vstack([tgt[:100], [[np.inf, 0, 0]]])), c.run(None, 0.1, 2).num_correspondences)[1])
attempt("zero iterations", lambda: (c.set_clouds_f64(src, tgt), c.run(None, 0.1, 0).iterations)[1])
attempt("ring forced, radius 1e30", lambda: (c.set_ring_search(1), c.set_clouds_f64(src, tgt), c.run(None, 1e30, 2).num_correspondences)[2])
attempt("ring forced, radius 1e-9", lambda: (c.set_ring_search(1), c.set_clouds_f64(src, tgt), c.run(None, 1e-9, 2).num_correspondences)[2])
print("done")
