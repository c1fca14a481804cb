"""ctypes bindings for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module, and only as the checker.  The product (visma_amd/) never
does.

Two libraries:
  * libvisma_oracle.so  -- our C restatement (oracle/icp_oracle.c), always built.
  * _ref/libvisma_ref.so -- the REAL reference path compiled from
    /root/reference by oracle/Makefile `ref` (present only where it was built;
    git-ignored; travels to the GPU box as a prebuilt file).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libvisma_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libvisma_ref.so")
REFERENCE_ROOT = "/root/reference"

NSTATS = 38
EST_POINT_TO_POINT = 1
EST_POINT_TO_PLANE = 2

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


def build(ref=True):
    """Compile the oracle; compile oracle/_ref too when the reference is here."""
    subprocess.check_call(["make", "-s", "-C", HERE])
    if ref and os.path.isdir(REFERENCE_ROOT):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


class VoResult(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("fitness", C.c_double),
                ("rmse", C.c_double), ("k", C.c_int64), ("iters", C.c_int32)]


class Result:
    def __init__(self, T, fitness, rmse, k, iters=None, idx=None, trace=None):
        self.T = np.array(T, dtype=np.float64).reshape(4, 4)
        self.fitness = float(fitness)
        self.rmse = float(rmse)
        self.k = int(k)
        self.iters = iters
        self.idx = idx
        self.trace = trace

    def __repr__(self):
        return "Result(k=%d fitness=%.6f rmse=%.6g iters=%s)" % (
            self.k, self.fitness, self.rmse, self.iters)


class Oracle:
    """The C restatement (groups A `vo_*` and B `vk_*` of icp_oracle.h)."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        self.lib = C.CDLL(ORACLE_SO)
        L = self.lib
        L.vo_nn_pass.restype = C.c_int64
        L.vo_nn_pass_grid.restype = C.c_int64
        L.vo_compute_rmse.restype = C.c_double
        L.vk_nn_pass_f32.restype = C.c_int64
        L.vk_nn_pass_f32_grid.restype = C.c_int64
        L.vo_num_threads.restype = C.c_int

    def num_threads(self):
        return int(self.lib.vo_num_threads())

    # ---- group A -------------------------------------------------------
    def transform_points(self, xyz, T):
        p = _f64(xyz, (-1, 3)).copy()
        T = _f64(T, (16,))
        self.lib.vo_transform_points(_ptr(p, _dp), C.c_int64(len(p)), _ptr(T, _dp))
        return p

    def transform_normals(self, nxyz, T):
        p = _f64(nxyz, (-1, 3)).copy()
        T = _f64(T, (16,))
        self.lib.vo_transform_normals(_ptr(p, _dp), C.c_int64(len(p)), _ptr(T, _dp))
        return p

    def nn_pass(self, src, tgt, max_dist, grid=False):
        src = _f64(src, (-1, 3)); tgt = _f64(tgt, (-1, 3))
        idx = np.empty(len(src), np.int32)
        d2 = np.empty(len(src), np.float64)
        e2 = C.c_double(0)
        fn = self.lib.vo_nn_pass_grid if grid else self.lib.vo_nn_pass
        k = fn(_ptr(src, _dp), C.c_int64(len(src)), _ptr(tgt, _dp),
               C.c_int64(len(tgt)), C.c_double(max_dist), _ptr(idx, _ip),
               _ptr(d2, _dp), C.byref(e2))
        return int(k), idx, d2, e2.value

    def nn_distance(self, src, tgt):
        src = _f64(src, (-1, 3)); tgt = _f64(tgt, (-1, 3))
        d = np.empty(len(src), np.float64)
        self.lib.vo_nn_distance(_ptr(src, _dp), C.c_int64(len(src)), _ptr(tgt, _dp),
                                C.c_int64(len(tgt)), _ptr(d, _dp))
        return d

    def compute_rmse(self, src, tgt, corr):
        src = _f64(src, (-1, 3)); tgt = _f64(tgt, (-1, 3))
        corr = np.ascontiguousarray(corr, np.int32).reshape(-1, 2)
        return float(self.lib.vo_compute_rmse(_ptr(src, _dp), _ptr(tgt, _dp),
                                              _ptr(corr, _ip), C.c_int64(len(corr))))

    def umeyama(self, src, tgt, corr, with_scaling=False):
        src = _f64(src, (-1, 3)); tgt = _f64(tgt, (-1, 3))
        corr = np.ascontiguousarray(corr, np.int32).reshape(-1, 2)
        T = np.empty(16)
        self.lib.vo_umeyama(_ptr(src, _dp), _ptr(tgt, _dp), _ptr(corr, _ip),
                            C.c_int64(len(corr)), C.c_int(int(with_scaling)),
                            _ptr(T, _dp))
        return T.reshape(4, 4)

    def jtj_jtr(self, src, tgt, corr, tgt_normals=None):
        src = _f64(src, (-1, 3)); tgt = _f64(tgt, (-1, 3))
        corr = np.ascontiguousarray(corr, np.int32).reshape(-1, 2)
        JTJ = np.empty(36); JTr = np.empty(6); r2 = C.c_double(0)
        if tgt_normals is None:
            self.lib.vo_jtj_jtr_point_to_point(
                _ptr(src, _dp), _ptr(tgt, _dp), _ptr(corr, _ip),
                C.c_int64(len(corr)), _ptr(JTJ, _dp), _ptr(JTr, _dp), C.byref(r2))
        else:
            n = _f64(tgt_normals, (-1, 3))
            self.lib.vo_jtj_jtr_point_to_plane(
                _ptr(src, _dp), _ptr(tgt, _dp), _ptr(n, _dp), _ptr(corr, _ip),
                C.c_int64(len(corr)), _ptr(JTJ, _dp), _ptr(JTr, _dp), C.byref(r2))
        return JTJ.reshape(6, 6), JTr, r2.value

    def solve_jacobian_system(self, JTJ, JTr):
        JTJ = _f64(JTJ, (36,)); JTr = _f64(JTr, (6,))
        T = np.empty(16)
        ok = self.lib.vo_solve_jacobian_system(_ptr(JTJ, _dp), _ptr(JTr, _dp), _ptr(T, _dp))
        return bool(ok), T.reshape(4, 4)

    def vector6d_to_matrix4d(self, x):
        x = _f64(x, (6,)); T = np.empty(16)
        self.lib.vo_vector6d_to_matrix4d(_ptr(x, _dp), _ptr(T, _dp))
        return T.reshape(4, 4)

    def point_to_plane_update(self, src, tgt, tgt_normals, corr):
        src = _f64(src, (-1, 3)); tgt = _f64(tgt, (-1, 3)); n = _f64(tgt_normals, (-1, 3))
        corr = np.ascontiguousarray(corr, np.int32).reshape(-1, 2)
        T = np.empty(16)
        self.lib.vo_point_to_plane_update(_ptr(src, _dp), _ptr(tgt, _dp), _ptr(n, _dp),
                                          _ptr(corr, _ip), C.c_int64(len(corr)), _ptr(T, _dp))
        return T.reshape(4, 4)

    def _icp(self, fn, src, tgt, max_dist, init, max_iter, rel_fitness, rel_rmse,
             with_scaling, grid, extra):
        src = _f64(src, (-1, 3)); tgt = _f64(tgt, (-1, 3))
        init = _f64(np.eye(4) if init is None else init, (16,))
        out = VoResult()
        idx = np.empty(len(src), np.int32)
        trace = np.full((max_iter + 1, 19), np.nan)
        rc = fn(src, tgt, init, out, idx, trace)
        res = Result(list(out.T), out.fitness, out.rmse, out.k, out.iters, idx,
                     trace[:out.iters + 1])
        res.rc = rc
        return res

    def registration_icp(self, src, tgt, max_dist, init=None, max_iter=30,
                         rel_fitness=1e-6, rel_rmse=1e-6, estimator=EST_POINT_TO_POINT,
                         with_scaling=False, tgt_normals=None, grid=True):
        n = None if tgt_normals is None else _f64(tgt_normals, (-1, 3))

        def call(s, t, i, out, idx, trace):
            return self.lib.vo_registration_icp(
                _ptr(s, _dp), C.c_int64(len(s)), _ptr(t, _dp), C.c_int64(len(t)),
                _ptr(n, _dp), C.c_double(max_dist), _ptr(i, _dp), C.c_int(estimator),
                C.c_int(int(with_scaling)), C.c_double(rel_fitness),
                C.c_double(rel_rmse), C.c_int(max_iter), C.c_int(int(grid)),
                C.byref(out), _ptr(idx, _ip), _ptr(trace, _dp))
        return self._icp(call, src, tgt, max_dist, init, max_iter, rel_fitness,
                         rel_rmse, with_scaling, grid, None)

    def register_model_to_scene(self, model, scene, level, max_dist, max_iter=30,
                                rel_fitness=1e-6, rel_rmse=1e-6, point_to_plane=False,
                                scene_normals=None):
        model = _f64(model, (-1, 3)); scene = _f64(scene, (-1, 3))
        n = None if scene_normals is None else _f64(scene_normals, (-1, 3))
        out = VoResult(); bl = C.c_int(-1)
        self.lib.vo_register_model_to_scene(
            _ptr(model, _dp), C.c_int64(len(model)), _ptr(scene, _dp),
            C.c_int64(len(scene)), _ptr(n, _dp), C.c_int(level), C.c_double(max_dist),
            C.c_int(int(point_to_plane)), C.c_double(rel_fitness), C.c_double(rel_rmse),
            C.c_int(max_iter), C.byref(out), C.byref(bl))
        r = Result(list(out.T), out.fitness, out.rmse, out.k, out.iters)
        r.best_level = bl.value
        return r

    # SO(3)/SE(3)
    def hat(self, u):
        u = _f64(u, (3,)); M = np.empty(9)
        self.lib.vo_hat(_ptr(u, _dp), _ptr(M, _dp))
        return M.reshape(3, 3)

    def rodrigues(self, w, jac=True):
        w = _f64(w, (3,)); R = np.empty(9); D = np.empty(27) if jac else None
        self.lib.vo_rodrigues(_ptr(w, _dp), _ptr(R, _dp), _ptr(D, _dp))
        return R.reshape(3, 3), (D.reshape(9, 3) if jac else None)

    def invrodrigues(self, R, jac=True):
        R = _f64(R, (9,)); w = np.empty(3); D = np.empty(27) if jac else None
        self.lib.vo_invrodrigues(_ptr(R, _dp), _ptr(w, _dp), _ptr(D, _dp))
        return w, (D.reshape(3, 9) if jac else None)

    def se3_compose(self, Ra, ta, Rb, tb):
        Ra = _f64(Ra, (9,)); ta = _f64(ta, (3,)); Rb = _f64(Rb, (9,)); tb = _f64(tb, (3,))
        R = np.empty(9); t = np.empty(3)
        self.lib.vo_se3_compose(_ptr(Ra, _dp), _ptr(ta, _dp), _ptr(Rb, _dp), _ptr(tb, _dp),
                                _ptr(R, _dp), _ptr(t, _dp))
        return R.reshape(3, 3), t

    def se3_act(self, R, t, v):
        R = _f64(R, (9,)); t = _f64(t, (3,)); v = _f64(v, (3,)); o = np.empty(3)
        self.lib.vo_se3_act(_ptr(R, _dp), _ptr(t, _dp), _ptr(v, _dp), _ptr(o, _dp))
        return o

    def se3_inv(self, R, t):
        R = _f64(R, (9,)); t = _f64(t, (3,)); Ri = np.empty(9); ti = np.empty(3)
        self.lib.vo_se3_inv(_ptr(R, _dp), _ptr(t, _dp), _ptr(Ri, _dp), _ptr(ti, _dp))
        return Ri.reshape(3, 3), ti

    def voxel_down_sample(self, xyz, voxel_size, normals=None, colors=None):
        """Returns (points, normals, colors) sorted by voxel index (ix, iy, iz)."""
        return _voxel(self.lib.vo_voxel_down_sample, xyz, voxel_size, normals, colors)

    def estimate_normals(self, xyz, knn=30, radius=None, normals=None):
        """open3d::EstimateNormals, neighbours by exhaustive scan (O(n^2): small clouds)."""
        return _normals(self.lib.vo_estimate_normals, xyz, knn, radius, normals)

    def sample_mesh(self, V, F, uniforms, quirks=True):
        V = _f64(V, (-1, 3)); F = np.ascontiguousarray(F, np.int32).reshape(-1, 3)
        u = _f64(uniforms, (-1, 3)); n = len(u)
        out = np.empty((max(n, 1), 3))
        self.lib.vo_sample_mesh.restype = C.c_int64
        m = self.lib.vo_sample_mesh(_ptr(V, _dp), _ptr(F, _ip), C.c_int64(len(F)), _ptr(u, _dp),
                                    C.c_int64(n), C.c_int(int(bool(quirks))), _ptr(out, _dp))
        return out[:m].copy()

    def point_mesh_sqdist(self, P, V, F):
        P = _f64(P, (-1, 3)); V = _f64(V, (-1, 3)); F = np.ascontiguousarray(F, np.int32).reshape(-1, 3)
        d2 = np.empty(len(P)); face = np.empty(len(P), np.int32); cl = np.empty((len(P), 3))
        self.lib.vo_point_mesh_sqdist(_ptr(P, _dp), C.c_int64(len(P)), _ptr(V, _dp), _ptr(F, _ip),
                                      C.c_int64(len(F)), _ptr(d2, _dp), _ptr(face, _ip), _ptr(cl, _dp))
        return d2, face, cl

    def error_metric(self, errors):
        e = _f64(errors, (-1,)); out = np.empty(5)
        self.lib.vo_error_metric(_ptr(e, _dp), C.c_int64(len(e)), _ptr(out, _dp))
        return dict(zip(("mean", "std", "median", "min", "max"), out))

    def svd3(self, A):
        A = _f64(A, (9,)); U = np.empty(9); s = np.empty(3); V = np.empty(9)
        self.lib.vo_svd3(_ptr(A, _dp), _ptr(U, _dp), _ptr(s, _dp), _ptr(V, _dp))
        return U.reshape(3, 3), s, V.reshape(3, 3)

    # ---- group B (kernel specification) --------------------------------
    def k_nn_pass(self, src32, tgt32, T32, r2f, grid=False):
        src32 = np.ascontiguousarray(src32, np.float32)
        tgt32 = np.ascontiguousarray(tgt32, np.float32)
        T32 = np.ascontiguousarray(T32, np.float32).reshape(12)
        ns, ss = src32.shape; nt, ts = tgt32.shape
        idx = np.empty(ns, np.int32); d2 = np.empty(ns, np.float32)
        fn = self.lib.vk_nn_pass_f32_grid if grid else self.lib.vk_nn_pass_f32
        k = fn(_ptr(src32, _fp), C.c_int64(ns), C.c_int(ss), _ptr(tgt32, _fp),
               C.c_int64(nt), C.c_int(ts), _ptr(T32, _fp), C.c_float(r2f),
               _ptr(idx, _ip), _ptr(d2, _fp))
        return int(k), idx, d2

    def k_reduce_stats(self, src32, tgt32, idx, T64, offset=None):
        src32 = np.ascontiguousarray(src32, np.float32)
        tgt32 = np.ascontiguousarray(tgt32, np.float32)
        idx = np.ascontiguousarray(idx, np.int32)
        T64 = _f64(T64, (-1,))[:12].copy()
        st = np.empty(NSTATS)
        self.lib.vk_reduce_stats(_ptr(src32, _fp), C.c_int64(src32.shape[0]),
                                 C.c_int(src32.shape[1]), _ptr(tgt32, _fp),
                                 C.c_int(tgt32.shape[1]), _ptr(idx, _ip),
                                 _ptr(T64, _dp),
                                 None if offset is None else _ptr(_f64(offset, (3,)), _dp),
                                 _ptr(st, _dp))
        return st

    def k_solve_kabsch(self, stats, with_scaling=False):
        st = _f64(stats, (NSTATS,)); T = np.empty(16)
        self.lib.vk_solve_kabsch_from_stats(_ptr(st, _dp), C.c_int(int(with_scaling)),
                                            _ptr(T, _dp))
        return T.reshape(4, 4)

    def k_solve_gn(self, stats):
        st = _f64(stats, (NSTATS,)); T = np.empty(16)
        ok = self.lib.vk_solve_gn_from_stats(_ptr(st, _dp), _ptr(T, _dp))
        return bool(ok), T.reshape(4, 4)

    def k_registration_icp(self, src, tgt, max_dist, init=None, max_iter=30,
                           rel_fitness=1e-6, rel_rmse=1e-6, with_scaling=False,
                           grid=True):
        def call(s, t, i, out, idx, trace):
            return self.lib.vk_registration_icp(
                _ptr(s, _dp), C.c_int64(len(s)), _ptr(t, _dp), C.c_int64(len(t)),
                C.c_double(max_dist), _ptr(i, _dp), C.c_int(int(with_scaling)),
                C.c_double(rel_fitness), C.c_double(rel_rmse), C.c_int(max_iter),
                C.c_int(int(grid)), C.byref(out), _ptr(idx, _ip), _ptr(trace, _dp))
        return self._icp(call, src, tgt, max_dist, init, max_iter, rel_fitness,
                         rel_rmse, with_scaling, grid, None)


def _normals(fn, xyz, knn, radius, normals):
    """radius None: KNN(knn); knn None: Radius(radius); both: Hybrid(radius, knn)."""
    p = _f64(xyz, (-1, 3)); n = len(p)
    nin = None if normals is None else _f64(normals, (-1, 3))
    kind = 0 if radius is None else (1 if knn is None else 2)
    out = np.empty((max(n, 1), 3))
    fn.restype = None
    fn(_ptr(p, _dp), C.c_int64(n), _ptr(nin, _dp), C.c_int(kind), C.c_int(int(knn or 0)),
       C.c_double(float(radius or 0.0)), _ptr(out, _dp))
    return out[:n].copy()


def _voxel(fn, xyz, voxel_size, normals, colors):
    fn.restype = C.c_int64
    p = _f64(xyz, (-1, 3)); n = len(p)
    nn = None if normals is None else _f64(normals, (-1, 3))
    cc = None if colors is None else _f64(colors, (-1, 3))
    op = np.empty((max(n, 1), 3)); on = np.empty((max(n, 1), 3)); oc = np.empty((max(n, 1), 3))
    m = fn(_ptr(p, _dp), _ptr(nn, _dp), _ptr(cc, _dp), C.c_int64(n), C.c_double(voxel_size),
           _ptr(op, _dp), _ptr(on, _dp), _ptr(oc, _dp))
    return (op[:m].copy(), None if nn is None else on[:m].copy(), None if cc is None else oc[:m].copy())


class Ref:
    """The real reference (Open3D 0.3.0 RegistrationICP et al.), when built."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO + " (run `make -C oracle ref` where "
                                    "/root/reference exists)")
        self.lib = C.CDLL(REF_SO)
        self.lib.ref_compute_rmse.restype = C.c_double

    def registration_icp(self, src, tgt, max_dist, init=None, max_iter=30,
                         rel_fitness=1e-6, rel_rmse=1e-6, estimator=EST_POINT_TO_POINT,
                         with_scaling=False, src_normals=None, tgt_normals=None):
        src = _f64(src, (-1, 3)); tgt = _f64(tgt, (-1, 3))
        sn = None if src_normals is None else _f64(src_normals, (-1, 3))
        tn = None if tgt_normals is None else _f64(tgt_normals, (-1, 3))
        init = _f64(np.eye(4) if init is None else init, (16,))
        T = np.empty(16); fit = C.c_double(); rmse = C.c_double(); k = C.c_int64()
        idx = np.empty(len(src), np.int32)
        self.lib.ref_registration_icp(
            _ptr(src, _dp), C.c_int64(len(src)), _ptr(sn, _dp), _ptr(tgt, _dp),
            C.c_int64(len(tgt)), _ptr(tn, _dp), C.c_double(max_dist), _ptr(init, _dp),
            C.c_int(estimator), C.c_int(int(with_scaling)), C.c_double(rel_fitness),
            C.c_double(rel_rmse), C.c_int(max_iter), _ptr(T, _dp), C.byref(fit),
            C.byref(rmse), _ptr(idx, _ip), C.byref(k))
        return Result(T, fit.value, rmse.value, k.value, None, idx)

    def evaluate_registration(self, src, tgt, max_dist, T=None):
        src = _f64(src, (-1, 3)); tgt = _f64(tgt, (-1, 3))
        T = _f64(np.eye(4) if T is None else T, (16,))
        fit = C.c_double(); rmse = C.c_double(); k = C.c_int64()
        idx = np.empty(len(src), np.int32)
        self.lib.ref_evaluate_registration(
            _ptr(src, _dp), C.c_int64(len(src)), _ptr(tgt, _dp), C.c_int64(len(tgt)),
            C.c_double(max_dist), _ptr(T, _dp), C.byref(fit), C.byref(rmse),
            _ptr(idx, _ip), C.byref(k))
        return Result(T, fit.value, rmse.value, k.value, None, idx)

    def transform_points(self, xyz, T, normals=None):
        p = _f64(xyz, (-1, 3)).copy()
        n = None if normals is None else _f64(normals, (-1, 3)).copy()
        T = _f64(T, (16,))
        self.lib.ref_transform_points(_ptr(p, _dp), C.c_int64(len(p)), _ptr(n, _dp),
                                      _ptr(T, _dp))
        return (p, n) if normals is not None else p

    def voxel_down_sample(self, xyz, voxel_size, normals=None, colors=None):
        """In the reference's own (hash-map) output order."""
        return _voxel(self.lib.ref_voxel_down_sample, xyz, voxel_size, normals, colors)

    def estimate_normals(self, xyz, knn=30, radius=None, normals=None):
        """open3d::EstimateNormals itself (KD-tree searches of KDTreeFlann)."""
        return _normals(self.lib.ref_estimate_normals, xyz, knn, radius, normals)

    def point_mesh_sqdist(self, P, V, F):
        """igl::AABB::squared_distance, as feh::MeasureSurfaceError calls it."""
        P = _f64(P, (-1, 3)); V = _f64(V, (-1, 3)); F = np.ascontiguousarray(F, np.int32).reshape(-1, 3)
        d2 = np.empty(len(P)); face = np.empty(len(P), np.int32); cl = np.empty((len(P), 3))
        self.lib.ref_igl_point_mesh_sqdist(_ptr(P, _dp), C.c_int64(len(P)), _ptr(V, _dp), C.c_int64(len(V)),
                                           _ptr(F, _ip), C.c_int64(len(F)), _ptr(d2, _dp), _ptr(face, _ip),
                                           _ptr(cl, _dp))
        return d2, face, cl

    # ---- gravity alignment of feh::AnnotationTool (oracle/ref_igl.cpp: the reference's expressions on the vendored Eigen)
    def find_plane_normal(self, xyz):
        xyz = _f64(xyz, (-1, 3)); out = np.empty(3)
        self.lib.ref_find_plane_normal(_ptr(xyz, _dp), C.c_int64(len(xyz)), _ptr(out, _dp))
        return out

    def jacobi_svd3(self, A):
        A = _f64(A, (9,)); U = np.empty(9); S = np.empty(3); V = np.empty(9)
        self.lib.ref_jacobi_svd3(_ptr(A, _dp), _ptr(U, _dp), _ptr(S, _dp), _ptr(V, _dp))
        return U.reshape(3, 3), S, V.reshape(3, 3)

    def rotation_between_vectors(self, u, v):
        u = _f64(u, (3,)); v = _f64(v, (3,)); R = np.empty(9)
        self.lib.ref_rotation_between_vectors(_ptr(u, _dp), _ptr(v, _dp), _ptr(R, _dp))
        return R.reshape(3, 3)

    # ---- file readers (open3d::ReadPointCloudFromPLY / ReadTriangleMeshFromPLY, igl::readOBJ) ----
    def _take(self, ptr, n, dtype):
        a = np.ctypeslib.as_array(ptr, shape=(max(n, 1),)).astype(dtype)[:n].copy()
        self.lib.ref_io_free(ptr)
        return a

    def read_ply_cloud(self, path):
        """-> dict(xyz, normals, colors) or None when the reference reader fails."""
        xyz, nrm, col = _dp(), _dp(), _dp()
        n, nn, nc = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self.lib.ref_io_free.argtypes = [C.c_void_p]
        if not self.lib.ref_read_ply_cloud(str(path).encode(), C.byref(xyz), C.byref(n), C.byref(nrm), C.byref(nn),
                                           C.byref(col), C.byref(nc)):
            return None
        return dict(xyz=self._take(xyz, 3 * n.value, np.float64).reshape(-1, 3),
                    normals=self._take(nrm, 3 * nn.value, np.float64).reshape(-1, 3),
                    colors=self._take(col, 3 * nc.value, np.float64).reshape(-1, 3))

    def read_pcd_cloud(self, path):
        """open3d::ReadPointCloudFromPCD -> dict(xyz, normals, colors) or None when the reference reader fails."""
        xyz, nrm, col = _dp(), _dp(), _dp()
        n, nn, nc = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self.lib.ref_io_free.argtypes = [C.c_void_p]
        if not self.lib.ref_read_pcd_cloud(str(path).encode(), C.byref(xyz), C.byref(n), C.byref(nrm), C.byref(nn),
                                           C.byref(col), C.byref(nc)):
            return None
        return dict(xyz=self._take(xyz, 3 * n.value, np.float64).reshape(-1, 3),
                    normals=self._take(nrm, 3 * nn.value, np.float64).reshape(-1, 3),
                    colors=self._take(col, 3 * nc.value, np.float64).reshape(-1, 3))

    def write_pcd(self, path, xyz, normals=None, colors=None, ascii=False, compressed=False):
        """open3d::WritePointCloudToPCD (the reference's writer: makes the binary_compressed fixtures)."""
        xyz = _f64(xyz, (-1, 3))
        nn = None if normals is None else _f64(normals, (-1, 3))
        cc = None if colors is None else _f64(colors, (-1, 3))
        self.lib.ref_write_pcd.argtypes = [C.c_char_p, _dp, C.c_int64, _dp, _dp, C.c_int, C.c_int]
        return bool(self.lib.ref_write_pcd(str(path).encode(), _ptr(xyz, _dp), len(xyz),
                                           None if nn is None else _ptr(nn, _dp), None if cc is None else _ptr(cc, _dp),
                                           int(ascii), int(compressed)))

    def read_ply_mesh(self, path):
        xyz, tri = _dp(), _ip()
        n, nt = C.c_int64(0), C.c_int64(0)
        self.lib.ref_io_free.argtypes = [C.c_void_p]
        if not self.lib.ref_read_ply_mesh(str(path).encode(), C.byref(xyz), C.byref(n), C.byref(tri), C.byref(nt)):
            return None
        return dict(xyz=self._take(xyz, 3 * n.value, np.float64).reshape(-1, 3),
                    faces=self._take(tri, 3 * nt.value, np.int32).reshape(-1, 3))

    def read_obj(self, path):
        """igl::readOBJ(path, V, F) -> (V [nv x vcols], F [nf x fcols]) or None."""
        V, F = _dp(), _ip()
        nv, nf, vc, fc = C.c_int64(0), C.c_int64(0), C.c_int(0), C.c_int(0)
        if not self.lib.ref_igl_read_obj(str(path).encode(), C.byref(V), C.byref(nv), C.byref(vc), C.byref(F),
                                         C.byref(nf), C.byref(fc)):
            return None
        self.lib.ref_io_free.argtypes = [C.c_void_p]
        v = self._take(V, nv.value * vc.value, np.float64).reshape(nv.value, max(vc.value, 0))
        f = self._take(F, nf.value * fc.value, np.int32).reshape(nf.value, max(fc.value, 0))
        return v, f

    def nn_distance(self, src, tgt):
        src = _f64(src, (-1, 3)); tgt = _f64(tgt, (-1, 3))
        d = np.empty(len(src))
        self.lib.ref_nn_distance(_ptr(src, _dp), C.c_int64(len(src)), _ptr(tgt, _dp),
                                 C.c_int64(len(tgt)), _ptr(d, _dp))
        return d

    def compute_rmse(self, src, tgt, corr, estimator=EST_POINT_TO_POINT, tgt_normals=None):
        src = _f64(src, (-1, 3)); tgt = _f64(tgt, (-1, 3))
        tn = None if tgt_normals is None else _f64(tgt_normals, (-1, 3))
        corr = np.ascontiguousarray(corr, np.int32).reshape(-1, 2)
        return float(self.lib.ref_compute_rmse(
            _ptr(src, _dp), C.c_int64(len(src)), _ptr(tgt, _dp), C.c_int64(len(tgt)),
            _ptr(tn, _dp), _ptr(corr, _ip), C.c_int64(len(corr)), C.c_int(estimator)))

    def compute_transformation(self, src, tgt, corr, estimator=EST_POINT_TO_POINT,
                               with_scaling=False, tgt_normals=None):
        src = _f64(src, (-1, 3)); tgt = _f64(tgt, (-1, 3))
        tn = None if tgt_normals is None else _f64(tgt_normals, (-1, 3))
        corr = np.ascontiguousarray(corr, np.int32).reshape(-1, 2)
        T = np.empty(16)
        self.lib.ref_compute_transformation(
            _ptr(src, _dp), C.c_int64(len(src)), _ptr(tgt, _dp), C.c_int64(len(tgt)),
            _ptr(tn, _dp), _ptr(corr, _ip), C.c_int64(len(corr)), C.c_int(estimator),
            C.c_int(int(with_scaling)), _ptr(T, _dp))
        return T.reshape(4, 4)

    def solve_jacobian_system(self, JTJ, JTr):
        JTJ = _f64(JTJ, (36,)); JTr = _f64(JTr, (6,)); T = np.empty(16)
        ok = self.lib.ref_solve_jacobian_system(_ptr(JTJ, _dp), _ptr(JTr, _dp), _ptr(T, _dp))
        return bool(ok), T.reshape(4, 4)

    def vector6d_to_matrix4d(self, x):
        x = _f64(x, (6,)); T = np.empty(16)
        self.lib.ref_vector6d_to_matrix4d(_ptr(x, _dp), _ptr(T, _dp))
        return T.reshape(4, 4)

    def hat(self, u):
        u = _f64(u, (3,)); M = np.empty(9)
        self.lib.ref_hat(_ptr(u, _dp), _ptr(M, _dp))
        return M.reshape(3, 3)

    def rodrigues(self, w, jac=True):
        w = _f64(w, (3,)); R = np.empty(9); D = np.empty(27) if jac else None
        self.lib.ref_rodrigues(_ptr(w, _dp), _ptr(R, _dp), _ptr(D, _dp))
        return R.reshape(3, 3), (D.reshape(9, 3) if jac else None)

    def invrodrigues(self, R, jac=True):
        R = _f64(R, (9,)); w = np.empty(3); D = np.empty(27) if jac else None
        self.lib.ref_invrodrigues(_ptr(R, _dp), _ptr(w, _dp), _ptr(D, _dp))
        return w, (D.reshape(3, 9) if jac else None)

    # ---- core/se3.h: the reference's own SE3Type / SO3Type (oracle/ref_se3.cpp); g = [R | t] row-major 3 x 4
    def se3_compose(self, a, b):
        a = _f64(a, (12,)); b = _f64(b, (12,)); o = np.empty(12)
        self.lib.ref_se3_compose(_ptr(a, _dp), _ptr(b, _dp), _ptr(o, _dp))
        return o.reshape(3, 4)

    def se3_act(self, g, v):
        g = _f64(g, (12,)); v = _f64(v, (3,)); o = np.empty(3)
        self.lib.ref_se3_act(_ptr(g, _dp), _ptr(v, _dp), _ptr(o, _dp))
        return o

    def se3_inv(self, g):
        g = _f64(g, (12,)); o = np.empty(12)
        self.lib.ref_se3_inv(_ptr(g, _dp), _ptr(o, _dp))
        return o.reshape(3, 4)

    def se3_matrix(self, g):
        g = _f64(g, (12,)); o = np.empty(16)
        self.lib.ref_se3_matrix(_ptr(g, _dp), _ptr(o, _dp))
        return o.reshape(4, 4)

    def so3_exp(self, w):
        w = _f64(w, (3,)); o = np.empty(9)
        self.lib.ref_so3_exp(_ptr(w, _dp), _ptr(o, _dp))
        return o.reshape(3, 3)

    def so3_log(self, R):
        R = _f64(R, (9,)); o = np.empty(3)
        self.lib.ref_so3_log(_ptr(R, _dp), _ptr(o, _dp))
        return o

    def so3_axis_angle(self, axis, angle):
        axis = _f64(axis, (3,)); o = np.empty(9)
        self.lib.ref_so3_axis_angle(_ptr(axis, _dp), C.c_double(float(angle)), _ptr(o, _dp))
        return o.reshape(3, 3)
