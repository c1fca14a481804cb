#!/usr/bin/env python3
"""Generate tests/golden/annotation.npz: ONE object of feh::AnnotationTool (src/annotation.cpp:71-168) end to end through the
reference's own code -- Eigen's JacobiSVD and Quaternion for the gravity alignment (oracle/ref_igl.cpp restates
include/geometry.h:18-26 and core/utils.h:229-233 on the vendored Eigen: those headers need OpenCV / jsoncpp / abseil
and cannot be compiled here), Open3D's VoxelDownSample, PointCloud::Transform and RegistrationICP for the rest --
plus known-answer vectors of the pieces (3 x 3 SVDs with their signs, plane normals, rotations between vectors).

The object: the reference's chair (misc/hermanmiller_aeron.obj) scanned from one side in a sensor frame that is tilted
against gravity, standing on a floor patch; the model cloud is the `2 x |scan|` surface samples the tool draws
(include/geometry.h:29-64 seeds from the wall clock: the samples are INPUT here, not reproduced).
T0's translation column and last row: the reference leaves them uninitialised (`Eigen::Matrix<double, 4, 4> T0;`
then only block<3,3> and (3,3) are set, src/annotation.cpp:87-89); the fixture takes them as zero -- what the
uninitialised stack holds when it holds nothing, and the only reading under which the tool's own output is meaningful.

    python tests/golden/gen_annotation.py        (this container only: needs oracle/_ref and /root/reference)"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from oracle.oracle import Ref  # noqa: E402


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], float)


def T4(R=None, t=None):
    T = np.eye(4)
    if R is not None:
        T[:3, :3] = R
    if t is not None:
        T[:3, 3] = t
    return T


def main():
    ref = Ref()
    rng = np.random.Generator(np.random.Philox(77))
    out = {}
    # ---- known answers of the pieces
    A = []
    for k in range(60):
        if k % 3 == 0:
            X = rng.standard_normal((40, 3)) * rng.uniform(0.01, 3, 3)
            X[:, rng.integers(3)] *= 1e-3
            A.append(np.cov((X @ np.linalg.qr(rng.standard_normal((3, 3)))[0]).T))
        elif k % 3 == 1:
            A.append(rng.standard_normal((3, 3)) * 10 ** rng.uniform(-6, 6))
        else:
            M = rng.standard_normal((3, 3)); M = M @ M.T; M[rng.integers(3)] *= 0
            A.append(M)
    A.append(np.zeros((3, 3))); A.append(np.eye(3)); A.append(np.diag([1.0, 1.0, 2.0]))
    svd = [ref.jacobi_svd3(a) for a in A]
    out.update(svd_A=np.array(A), svd_U=np.array([s[0] for s in svd]), svd_S=np.array([s[1] for s in svd]),
               svd_V=np.array([s[2] for s in svd]))
    uv = rng.standard_normal((40, 2, 3))
    uv[0, 1] = uv[0, 0] * 2.0                    # parallel
    uv[1] = [[0.02, 0.999, -0.01], [0, 1, 0]]    # a floor normal that is almost +Y already
    out.update(rot_uv=uv, rot_R=np.array([ref.rotation_between_vectors(u, v) for u, v in uv]))
    planes, normals = [], []
    for k in range(8):
        n = rng.standard_normal(3); n /= np.linalg.norm(n)
        P = rng.standard_normal((1500, 3)) * [1.0, 0.6, 1.0]
        P -= np.outer(P @ n, n)
        P += np.outer(rng.standard_normal(1500) * 0.004, n) + rng.standard_normal(3) * 3
        P = gg.f32(P).astype(np.float64)
        planes.append(P); normals.append(ref.find_plane_normal(P))
    out.update(plane_pts=np.array(planes), plane_n=np.array(normals))

    # ---- one object, end to end
    V, F = gg.load_obj(gg.REF + "/misc/hermanmiller_aeron.obj")
    # world (gravity) frame: y up; the chair stands on the floor y = 0, turned by 140 degrees, somewhere in the room
    dense = gg.sample_mesh(V, F, 60000, 31)
    dense[:, 1] -= dense[:, 1].min()
    T_world_model = T4(gg.rot_y(math.radians(140.0)), [1.3, 0.0, -0.7])
    dense_w = dense @ T_world_model[:3, :3].T + T_world_model[:3, 3]
    view = np.array([math.cos(2.2), 0.35, math.sin(2.2)])
    keep = np.argsort(-((dense_w - dense_w.mean(0)) @ view))[:51000]            # (a scan from several stations: most of the surface)
    keep = keep[rng.permutation(len(keep))[:16000]]
    keep.sort()
    scan_w = dense_w[keep] + rng.standard_normal((len(keep), 3)) * 0.002
    floor_w = np.stack([rng.uniform(-0.5, 3.0, 6000), rng.standard_normal(6000) * 0.003, rng.uniform(-2.5, 1.0, 6000)], 1)
    # the sensor frame: tilted against gravity and displaced
    T_sensor_world = T4(rot_x(math.radians(23.0)) @ gg.rot_y(math.radians(-35.0)), [0.4, 1.1, 2.0])
    to_sensor = lambda p: gg.f32(p @ T_sensor_world[:3, :3].T + T_sensor_world[:3, 3]).astype(np.float64)   # noqa: E731
    scan_raw, floor = to_sensor(scan_w), to_sensor(floor_w)
    voxel, level, thr = 0.01, 24, 0.02                                   # cfg/tool.json:19-22
    # src/annotation.cpp:82-91
    floor_n = ref.find_plane_normal(floor)
    R = ref.rotation_between_vectors(floor_n, [0.0, 1.0, 0.0])
    T0 = T4(R)
    # :112-119
    scan = ref.voxel_down_sample(scan_raw, voxel)[0]
    scan = ref.transform_points(scan, T0)
    mean = scan.mean(0)
    T1 = T4(None, [-mean[0], -scan[:, 1].min(), -mean[2]])
    scan = ref.transform_points(scan, T1)
    # :121-132  (the samples are an input: wall-clock seed)
    model_pts = gg.f32(gg.sample_mesh(V, F, 2 * len(scan), 32)).astype(np.float64)
    mean = model_pts.mean(0)
    T2 = T4(None, [-mean[0], -model_pts[:, 1].min(), -mean[2]])
    model = ref.transform_points(model_pts, T2)
    # :144 RegisterModelToScene (:29-64)
    ks, Ts = [], []
    for i in range(level):
        r = ref.registration_icp(model, scan, thr, init=T4(gg.rot_y(2 * math.pi / level * i)), max_iter=30,
                                 rel_fitness=1e-6, rel_rmse=1e-6)
        ks.append(r.k); Ts.append(np.asarray(r.T).reshape(4, 4))
    best = int(np.argmax(ks))                                             # first maximum = "strictly more"
    T3 = Ts[best]
    # :147-153
    A10 = T1 @ T0
    inv = A10.copy()
    inv[:3, :3] = A10[:3, :3].T
    inv[:3, 3] = -inv[:3, :3] @ A10[:3, 3]
    Ttot = inv @ T3 @ T2
    # sanity: the model lands on the scan
    moved = model_pts @ Ttot[:3, :3].T + Ttot[:3, 3]
    e = ref.evaluate_registration(moved, scan_raw, thr)
    print("object: |scan| %d -> %d voxels, model %d, best level %d of K %s, K of Ttot on the raw scan %d (fitness %.3f)"
          % (len(scan_raw), len(scan), len(model_pts), best, ks, e.k, e.fitness))
    out.update(floor=gg.f32(floor), scan_raw=gg.f32(scan_raw), model_pts=gg.f32(model_pts), voxel=voxel, level=level,
               threshold=thr, floor_n=floor_n, T0=T0, T1=T1, T2=T2, T3=T3, Ttot=Ttot, best=best, sweep_k=np.array(ks),
               n_scan_voxels=len(scan))
    np.savez_compressed(os.path.join(HERE, "annotation.npz"), **out)
    print("wrote annotation.npz")


if __name__ == "__main__":
    main()
