"""ctypes binding of the C ABI in include/visma_icp.h (libvisma_icp.so).

The library is the product: HIP kernels for gfx950 + the C++ host driver.
There is no Python or CPU implementation behind it -- if the shared library
is missing, or no GPU is visible when a context is created, this module
raises; it never falls back.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libvisma_icp.so")

NSTATS = 38
UNIQUE_ID_BYTES = 128
IPC_HANDLE_BYTES = 64
SOLVER_KABSCH, SOLVER_GN_EULER, SOLVER_GN_EXPMAP = 0, 1, 2
NN_AUTO, NN_BRUTE, NN_GRID = 0, 1, 2
OK = 0
ERR_NAMES = {1: "INVALID", 2: "NO_DEVICE", 3: "HIP", 4: "RCCL", 5: "STATE", 6: "ENGINE"}

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class IcpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("visma_icp error %d (%s): %s" % (code, ERR_NAMES.get(code, "?"), msg))
        self.code = code


class CResult(C.Structure):
    _fields_ = [("transformation", C.c_double * 16), ("fitness", C.c_double),
                ("inlier_rmse", C.c_double), ("num_correspondences", C.c_int64),
                ("iterations", C.c_int32), ("nn_passes", C.c_int32)]


class CTiming(C.Structure):
    _fields_ = [("nn_ms", C.c_double), ("nn_launches", C.c_int64),
                ("reduce_ms", C.c_double), ("reduce_launches", C.c_int64),
                ("aux_ms", C.c_double), ("aux_launches", C.c_int64),
                ("grid_candidates", C.c_double), ("grid_candidates_27cell", C.c_double),
                ("tile_workgroups", C.c_double), ("tile_fallback_workgroups", C.c_double), ("tile_parts", C.c_double),
                ("tile_points", C.c_double), ("tile_rows", C.c_double), ("f64_reranks", C.c_double),
                ("tile_phase_cycles", C.c_double * 7), ("grid_certified", C.c_double),
                ("persist_launches", C.c_double), ("persist_passes", C.c_double), ("persist_ms", C.c_double),
                ("persist_aborts", C.c_double)]


class CPersistentInfo(C.Structure):
    _fields_ = [("struct_size", C.c_int), ("enabled", C.c_int), ("last_loop_persistent", C.c_int), ("last_loop_passes", C.c_int),
                ("launches", C.c_double), ("passes", C.c_double), ("aborts", C.c_double), ("timeout_ms", C.c_double),
                ("cu_share", C.c_double), ("device_slots", C.c_int), ("reserved", C.c_int)]


class CProblem(C.Structure):
    _fields_ = [("src_xyz", _dp), ("ns", C.c_int64), ("tgt_xyz", _dp), ("nt", C.c_int64),
                ("init", C.c_double * 16), ("max_dist", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, C.c_int)
MINREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_int64)
ENG_SET = C.CFUNCTYPE(C.c_int, C.c_void_p, _fp, C.c_int64)
ENG_NN = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, C.c_double)
ENG_REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, _dp, C.c_int, _dp)
ENG_CORR = C.CFUNCTYPE(C.c_int, C.c_void_p, _ip, _fp)


class CMeshSource(C.Structure):
    _fields_ = [("V", C.POINTER(C.c_double)), ("nv", C.c_int64), ("F", C.POINTER(C.c_int32)), ("nf", C.c_int64),
                ("samples", C.c_int64), ("model_to_scene", C.POINTER(C.c_double))]


class CCorpusItem(C.Structure):
    _fields_ = [("model_xyz", C.POINTER(C.c_double)), ("n_model", C.c_int64),
                ("scene_xyz", C.POINTER(C.c_double)), ("n_scene", C.c_int64)]


class CCorpusParams(C.Structure):
    _fields_ = [("level", C.c_int), ("max_dist", C.c_double), ("max_iter", C.c_int), ("rel_fitness", C.c_double),
                ("rel_rmse", C.c_double), ("solver", C.c_int), ("chunk", C.c_int)]


class CCorpusResult(C.Structure):
    _fields_ = [("best", CResult), ("best_level", C.c_int32), ("device", C.c_int32), ("iterations_all_starts", C.c_int64)]


class CEngine(C.Structure):
    _fields_ = [("set_source", ENG_SET), ("set_target", ENG_SET),
                ("set_target_normals", ENG_SET), ("nn_pass", ENG_NN),
                ("reduce", ENG_REDUCE), ("get_correspondences", ENG_CORR)]


_lib = None


_seams = None


def load_seams():
    """The side build with the test seam visma_icp_create_with_engine (and the experiments that lost): what the
    CPU suite drives the host loop through.  Never the product library."""
    global _seams
    if _seams is None:
        from . import build
        _seams = _bind(C.CDLL(build.build_experiments()))
    return _seams


def load():
    """Load libvisma_icp.so (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    lib_path = os.environ.get("VISMA_ICP_LIB", LIB_PATH)     # (experiments: an alternative build of the same library)
    if not os.path.exists(lib_path):
        raise ImportError(
            "%s is missing: build it with `python -m visma_amd.build` (hipcc, gfx950). "
            "visma_amd has no non-HIP implementation." % lib_path)
    _lib = _bind(C.CDLL(lib_path))
    return _lib


def _bind(L):
    L.visma_icp_last_error.restype = C.c_char_p
    L.visma_icp_last_error.argtypes = [C.c_void_p]
    L.visma_icp_version.restype = C.c_char_p
    L.visma_icp_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    if hasattr(L, "visma_icp_create_with_engine"):
        L.visma_icp_create_with_engine.argtypes = [C.POINTER(C.c_void_p), C.POINTER(CEngine), C.c_void_p]
    L.visma_icp_destroy.argtypes = [C.c_void_p]
    L.visma_icp_set_clouds_f64.argtypes = [C.c_void_p, _dp, C.c_int64, C.c_int, _dp, C.c_int64, C.c_int]
    L.visma_icp_set_clouds_f64_voxel_target.argtypes = [C.c_void_p, _dp, C.c_int64, C.c_int, _dp, C.c_int64, C.c_int,
                                                        C.c_double, C.POINTER(C.c_int64)]
    L.visma_icp_get_voxel_target.argtypes = [C.c_void_p, _dp, C.c_int64]
    L.visma_icp_set_clouds_meshes_f64.argtypes = [C.c_void_p, C.POINTER(CMeshSource), C.c_int, C.c_int, C.c_uint64, _dp,
                                                  C.c_int64, C.c_int, C.c_double, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.visma_icp_get_mesh_source.argtypes = [C.c_void_p, _dp, C.c_int64]
    L.visma_icp_set_radius_hint.argtypes = [C.c_void_p, C.c_double]
    L.visma_icp_set_target.argtypes = [C.c_void_p, _fp, C.c_int64, C.c_int]
    L.visma_icp_set_source.argtypes = [C.c_void_p, _fp, C.c_int64, C.c_int]
    L.visma_icp_set_target_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.visma_icp_set_source_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.visma_icp_set_target_normals_f64.argtypes = [C.c_void_p, _dp, C.c_int64, C.c_int]
    L.visma_icp_get_search_kernel_used.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.visma_icp_forget_winners.argtypes = [C.c_void_p]
    L.visma_icp_run_batch_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(CProblem), C.c_int, C.c_int, C.c_double,
                                            C.c_double, C.c_int, C.POINTER(CResult), C.c_char_p, C.c_size_t]
    L.visma_icp_run_corpus.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(CCorpusItem), C.c_int64,
                                       C.POINTER(CCorpusParams), C.POINTER(C.c_int64), C.POINTER(CCorpusResult),
                                       C.c_char_p, C.c_size_t]
    L.visma_icp_nn_pass.argtypes = [C.c_void_p, _dp, C.c_double]
    L.visma_icp_reduce.argtypes = [C.c_void_p, _dp]
    L.visma_icp_get_correspondences.argtypes = [C.c_void_p, _ip, _ip, _fp, C.POINTER(C.c_int64)]
    L.visma_icp_solve_from_stats.argtypes = [_dp, C.c_int, C.c_int, _dp]
    L.visma_icp_run.argtypes = [C.c_void_p, _dp, C.c_double, C.c_int, C.c_double, C.c_double,
                                C.c_int, C.c_int, C.POINTER(CResult)]
    L.visma_icp_iterate.argtypes = [C.c_void_p, _dp, C.c_double, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(CResult)]
    L.visma_icp_run_point_to_plane.argtypes = [C.c_void_p, _dp, C.c_double, C.c_int, C.c_double,
                                               C.c_double, C.POINTER(CResult)]
    L.visma_icp_run_yaw_sweep.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double,
                                          C.c_double, C.c_int, C.POINTER(CResult),
                                          C.POINTER(C.c_int), C.POINTER(CResult)]
    L.visma_icp_estimate_normals.argtypes = [C.c_void_p, _dp, C.c_int64, _dp, C.c_int, C.c_int, C.c_double, _dp]
    L.visma_icp_run_batch_point_to_plane.argtypes = [C.c_void_p, C.POINTER(CProblem), C.POINTER(_dp), C.c_int, C.c_int,
                                                     C.c_double, C.c_double, C.POINTER(CResult)]
    L.visma_icp_run_batch.argtypes = [C.c_void_p, C.POINTER(CProblem), C.c_int, C.c_int,
                                      C.c_double, C.c_double, C.c_int, C.POINTER(CResult)]
    L.visma_icp_voxel_down_sample.argtypes = [C.c_void_p, _dp, C.c_int64, _dp, _dp, C.c_double, _dp, _dp,
                                              _dp, C.POINTER(C.c_int64)]
    L.visma_icp_sample_mesh.argtypes = [C.c_void_p, _dp, C.c_int64, _ip, C.c_int64, C.c_int64, C.c_int,
                                        C.c_uint64, _dp, _dp, C.POINTER(C.c_int64)]
    L.visma_icp_point_mesh_distance.argtypes = [C.c_void_p, _dp, C.c_int64, _dp, C.c_int64, _ip, C.c_int64,
                                                _dp, _ip, _dp]
    L.visma_icp_last_mesh_kernel_ms.argtypes = [C.c_void_p, _dp, _dp]
    L.visma_icp_set_mesh_search.argtypes = [C.c_void_p, C.c_int]
    L.visma_icp_error_metric.argtypes = [_dp, C.c_int64, _dp]
    L.visma_icp_measure_surface_error.argtypes = [C.c_void_p, _dp, C.c_int64, _ip, C.c_int64, _dp, C.c_int64,
                                                  _ip, C.c_int64, C.c_int64, C.c_int, C.c_uint64, _dp]
    L.visma_icp_set_target_shard.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _dp]
    L.visma_icp_set_minreduce.argtypes = [C.c_void_p, MINREDUCE_FN, C.c_void_p]
    L.visma_icp_set_search_precision.argtypes = [C.c_void_p, C.c_int]
    L.visma_icp_get_search_precision_used.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.visma_icp_set_nn_mode.argtypes = [C.c_void_p, C.c_int]
    L.visma_icp_get_nn_mode_used.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.visma_icp_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.visma_icp_set_device_loop.argtypes = [C.c_void_p, C.c_int]
    L.visma_icp_set_persistent.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.visma_icp_test_stall_command.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.visma_icp_get_sweep_info.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if hasattr(L, "visma_icp_set_ring_search"):          # (A/B runs load older builds through VISMA_ICP_LIB)
        L.visma_icp_set_ring_search.argtypes = [C.c_void_p, C.c_int]
        L.visma_icp_get_ring_search.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.visma_icp_set_persistent_cu_share.argtypes = [C.c_double]
    L.visma_icp_get_persistent_info.argtypes = [C.c_void_p, C.POINTER(CPersistentInfo)]
    L.visma_icp_get_timing_sized.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.visma_icp_get_timing.argtypes = [C.c_void_p, C.POINTER(CTiming), C.c_int]
    L.visma_icp_get_tile_config.argtypes = [C.POINTER(C.c_int)] * 3
    L.visma_icp_get_launch_config.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.visma_icp_comm_unique_id.argtypes = [C.c_void_p]
    L.visma_icp_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.visma_icp_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_int, C.c_int]
    L.visma_so3_rodrigues.argtypes = [_dp, _dp, _dp]
    L.visma_so3_invrodrigues.argtypes = [_dp, _dp, _dp]
    L.visma_so3_project.argtypes = [_dp, _dp]
    L.visma_so3_matrix_derivatives.argtypes = [_dp] * 7
    L.visma_icp_selftest_so3_jac.argtypes = [_dp, C.c_int, _dp, _dp, _dp, _dp, _dp]
    L.visma_icp_run_yaw_sweep_point_to_plane.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double,
                                                         C.c_double, C.POINTER(CResult), C.POINTER(C.c_int),
                                                         C.POINTER(CResult)]
    L.visma_icp_comm_ipc_export.argtypes = [C.c_void_p, C.c_void_p]
    L.visma_icp_comm_ipc_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.visma_icp_set_global_source_count.argtypes = [C.c_void_p, C.c_int64]
    return L


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if shape is None else a.reshape(shape)


def _p(a, t):
    return a.ctypes.data_as(t)


class Result:
    """Mirror of open3d::RegistrationResult (+ iteration counts)."""

    def __init__(self, c):
        self.transformation_ = np.array(list(c.transformation), dtype=np.float64).reshape(4, 4)
        self.fitness_ = float(c.fitness)
        self.inlier_rmse_ = float(c.inlier_rmse)
        self.num_correspondences = int(c.num_correspondences)
        self.iterations = int(c.iterations)
        self.nn_passes = int(c.nn_passes)
        self.correspondence_set_ = None

    def __repr__(self):
        return ("RegistrationResult(fitness=%.6f, inlier_rmse=%.6g, K=%d, iterations=%d)" %
                (self.fitness_, self.inlier_rmse_, self.num_correspondences, self.iterations))


class Context:
    """One ICP context = one HIP stream on one GPU (visma_icp_ctx)."""

    def __init__(self, device=0, engine=None):
        self.L = load() if engine is None else load_seams()
        self._h = C.c_void_p()
        self._keep = []
        if engine is None:
            rc = self.L.visma_icp_create(C.byref(self._h), int(device))
        else:
            self._keep.append(engine)
            rc = self.L.visma_icp_create_with_engine(C.byref(self._h), C.byref(engine), None)
        if rc != OK:
            msg = self.L.visma_icp_last_error(None)
            raise IcpError(rc, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.L.visma_icp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != OK:
            msg = self.L.visma_icp_last_error(self._h)
            raise IcpError(rc, msg.decode() if msg else "")

    # ---- clouds ----
    def set_clouds_f64(self, src, tgt):
        src = _f64(src, (-1, 3)); tgt = _f64(tgt, (-1, 3))
        self._chk(self.L.visma_icp_set_clouds_f64(self._h, _p(src, _dp), len(src), 3,
                                                  _p(tgt, _dp), len(tgt), 3))
        self.ns, self.nt = len(src), len(tgt)

    def set_clouds_f64_voxel_target(self, src, scene, voxel_size):
        """target = VoxelDownSample(scene, voxel_size), made and installed on the device -> its point count."""
        s = _f64(src, (-1, 3)); t = _f64(scene, (-1, 3))
        nt = C.c_int64(0)
        self._chk(self.L.visma_icp_set_clouds_f64_voxel_target(self._h, _p(s, _dp), len(s), 3, _p(t, _dp), len(t), 3,
                                                               float(voxel_size), C.byref(nt)))
        self.ns, self.nt = len(s), int(nt.value)
        return int(nt.value)

    def set_clouds_meshes_f64(self, meshes, scene, voxel_size=0.0, reference_quirks=0, seed=0):
        """meshes: [(V (nv,3) f64, F (nf,3) i32, samples, T (4,4) or None)]; the source is sampled, moved and
        concatenated on the device (feh::ICPRefinement's scene_est), the scene optionally voxel-down-sampled there.
        Returns (ns, nt)."""
        keep = []
        arr = (CMeshSource * max(len(meshes), 1))()
        for k, (V, F, n, T) in enumerate(meshes):
            V = np.ascontiguousarray(V, np.float64); F = np.ascontiguousarray(F, np.int32)
            Tm = None if T is None else np.ascontiguousarray(T, np.float64).reshape(16)
            keep += [V, F, Tm]
            arr[k].V = _p(V, _dp); arr[k].nv = len(V)
            arr[k].F = _p(F, C.POINTER(C.c_int32)); arr[k].nf = len(F)
            arr[k].samples = int(n)
            arr[k].model_to_scene = _p(Tm, _dp) if Tm is not None else None
        scene = np.ascontiguousarray(scene, np.float64)
        ns, nt = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.visma_icp_set_clouds_meshes_f64(self._h, arr, len(meshes), int(reference_quirks), int(seed),
                                                         _p(scene, _dp), len(scene), 3, float(voxel_size),
                                                         C.byref(ns), C.byref(nt)))
        self.ns, self.nt = int(ns.value), int(nt.value)
        return self.ns, self.nt

    def set_radius_hint(self, r):
        """The search radius of the next registration: the next upload builds the grid while it stages the source."""
        self._chk(self.L.visma_icp_set_radius_hint(self._h, float(r)))

    def get_mesh_source(self, ns):
        out = np.empty((int(ns), 3), np.float64)
        self._chk(self.L.visma_icp_get_mesh_source(self._h, _p(out, _dp), int(ns)))
        return out

    def get_voxel_target(self, nt):
        out = np.empty((max(nt, 1), 3))
        self._chk(self.L.visma_icp_get_voxel_target(self._h, _p(out, _dp), int(nt)))
        return out[:nt].copy()

    def set_target(self, xyz):
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        self._chk(self.L.visma_icp_set_target(self._h, _p(a, _fp), a.shape[0], a.shape[1]))
        self.nt = a.shape[0]

    def set_source(self, xyz):
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        self._chk(self.L.visma_icp_set_source(self._h, _p(a, _fp), a.shape[0], a.shape[1]))
        self.ns = a.shape[0]

    def set_target_device(self, ptr, n):
        self._chk(self.L.visma_icp_set_target_device(self._h, C.c_void_p(ptr), n))
        self.nt = n

    def set_source_device(self, ptr, n):
        self._chk(self.L.visma_icp_set_source_device(self._h, C.c_void_p(ptr), n))
        self.ns = n

    def set_target_normals_f64(self, n):
        n = _f64(n, (-1, 3))
        self._chk(self.L.visma_icp_set_target_normals_f64(self._h, _p(n, _dp), len(n), 3))

    # ---- kernels ----
    def nn_pass(self, T, max_dist):
        T = _f64(T, (16,))
        self._chk(self.L.visma_icp_nn_pass(self._h, _p(T, _dp), float(max_dist)))

    def reduce(self):
        st = np.empty(NSTATS)
        self._chk(self.L.visma_icp_reduce(self._h, _p(st, _dp)))
        return st

    def get_correspondences(self):
        si = np.empty(max(self.ns, 1), np.int32); ti = np.empty(max(self.ns, 1), np.int32)
        d2 = np.empty(max(self.ns, 1), np.float32); k = C.c_int64(0)
        self._chk(self.L.visma_icp_get_correspondences(self._h, _p(si, _ip), _p(ti, _ip),
                                                       _p(d2, _fp), C.byref(k)))
        k = k.value
        return si[:k].copy(), ti[:k].copy(), d2[:k].copy()

    def correspondence_index(self):
        """Per-source-point target index (-1 = none), like the oracle's idx."""
        si, ti, _ = self.get_correspondences()
        idx = np.full(self.ns, -1, np.int32)
        idx[si] = ti
        return idx

    # ---- loops ----
    def run(self, init=None, max_dist=0.05, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6,
            solver=SOLVER_KABSCH, with_scaling=False):
        init = _f64(np.eye(4) if init is None else init, (16,))
        out = CResult()
        self._chk(self.L.visma_icp_run(self._h, _p(init, _dp), float(max_dist), int(max_iter),
                                       float(rel_fitness), float(rel_rmse), int(solver),
                                       int(bool(with_scaling)), C.byref(out)))
        return Result(out)

    def iterate(self, T, max_dist, steps, solver=SOLVER_KABSCH, with_scaling=False):
        """Exactly `steps` fixed iterations from T; returns (T_new, Result of last pass)."""
        T = _f64(np.eye(4) if T is None else T, (16,)).copy()
        out = CResult()
        self._chk(self.L.visma_icp_iterate(self._h, _p(T, _dp), float(max_dist), int(steps),
                                           int(solver), int(bool(with_scaling)), C.byref(out)))
        return T.reshape(4, 4), Result(out)

    def run_point_to_plane(self, init=None, max_dist=0.05, max_iter=30, rel_fitness=1e-6,
                           rel_rmse=1e-6):
        init = _f64(np.eye(4) if init is None else init, (16,))
        out = CResult()
        self._chk(self.L.visma_icp_run_point_to_plane(self._h, _p(init, _dp), float(max_dist),
                                                      int(max_iter), float(rel_fitness),
                                                      float(rel_rmse), C.byref(out)))
        return Result(out)

    def run_yaw_sweep(self, level, max_dist, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6,
                      solver=SOLVER_KABSCH):
        best = CResult(); bl = C.c_int(-1); per = (CResult * level)()
        self._chk(self.L.visma_icp_run_yaw_sweep(self._h, int(level), float(max_dist),
                                                 int(max_iter), float(rel_fitness),
                                                 float(rel_rmse), int(solver), C.byref(best),
                                                 C.byref(bl), per))
        return Result(best), bl.value, [Result(p) for p in per]

    def run_yaw_sweep_point_to_plane(self, level, max_dist, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
        best = CResult(); bl = C.c_int(-1); per = (CResult * level)()
        self._chk(self.L.visma_icp_run_yaw_sweep_point_to_plane(self._h, int(level), float(max_dist), int(max_iter),
                                                                float(rel_fitness), float(rel_rmse), C.byref(best),
                                                                C.byref(bl), per))
        return Result(best), bl.value, [Result(p) for p in per]

    @staticmethod
    def make_batch(problems):
        """The visma_icp_problem array of (src, tgt, init, radius) tuples, built once: a caller that runs the
        same batch again (bench.py) does not pay the ctypes marshalling inside its timed region."""
        n = len(problems)
        arr = (CProblem * max(n, 1))(); keep = []
        for i, (src, tgt, init, r) in enumerate(problems):
            s = _f64(src, (-1, 3)); t = _f64(tgt, (-1, 3)); keep += [s, t]
            arr[i].src_xyz = _p(s, _dp); arr[i].ns = len(s)
            arr[i].tgt_xyz = _p(t, _dp); arr[i].nt = len(t)
            arr[i].init = (C.c_double * 16)(*_f64(np.eye(4) if init is None else init, (16,)))
            arr[i].max_dist = float(r)
        return (arr, n, keep, (CResult * max(n, 1))())

    def run_batch(self, problems, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6,
                  solver=SOLVER_KABSCH):
        """problems: (src, tgt, init, radius) tuples, or the value of make_batch()."""
        arr, n, _keep, out = problems if (isinstance(problems, tuple) and len(problems) == 4 and
                                          isinstance(problems[1], int)) else self.make_batch(problems)
        self._chk(self.L.visma_icp_run_batch(self._h, arr, n, int(max_iter), float(rel_fitness),
                                             float(rel_rmse), int(solver), out))
        return [Result(out[i]) for i in range(n)]

    def run_batch_point_to_plane(self, problems, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
        """problems: (src, tgt, tgt_normals or None, init, radius) -- the point-to-plane estimator, all in flight."""
        n = len(problems)
        arr = (CProblem * max(n, 1))(); nrm = (_dp * max(n, 1))(); keep = []
        for i, (src, tgt, normals, init, r) in enumerate(problems):
            s = _f64(src, (-1, 3)); t = _f64(tgt, (-1, 3)); keep += [s, t]
            arr[i].src_xyz = _p(s, _dp); arr[i].ns = len(s)
            arr[i].tgt_xyz = _p(t, _dp); arr[i].nt = len(t)
            arr[i].init = (C.c_double * 16)(*_f64(np.eye(4) if init is None else init, (16,)))
            arr[i].max_dist = float(r)
            if normals is not None:
                q = _f64(normals, (-1, 3)); keep.append(q)
                assert len(q) == len(t)
                nrm[i] = _p(q, _dp)
        out = (CResult * max(n, 1))()
        self._chk(self.L.visma_icp_run_batch_point_to_plane(self._h, arr, nrm, n, int(max_iter), float(rel_fitness),
                                                            float(rel_rmse), out))
        return [Result(out[i]) for i in range(n)]

    def estimate_normals(self, xyz, knn=30, radius=None, normals=None):
        """open3d::EstimateNormals on the GPU.  radius None: KNN(knn); knn None: Radius(radius); both: Hybrid."""
        p = _f64(xyz, (-1, 3)); n = len(p)
        nin = None if normals is None else _f64(normals, (-1, 3))
        kind = 0 if radius is None else (1 if knn is None else 2)
        out = np.empty((max(n, 1), 3))
        self._chk(self.L.visma_icp_estimate_normals(self._h, _p(p, _dp), n, None if nin is None else _p(nin, _dp),
                                                    kind, int(knn or 0), float(radius or 0.0), _p(out, _dp)))
        return out[:n].copy()

    def voxel_down_sample(self, xyz, voxel_size, normals=None, colors=None):
        """open3d::VoxelDownSample on the GPU -> (points, normals, colors), voxels in ascending index order."""
        p = _f64(xyz, (-1, 3)); n = len(p)
        nn = None if normals is None else _f64(normals, (-1, 3))
        cc = None if colors is None else _f64(colors, (-1, 3))
        op = np.empty((max(n, 1), 3)); on = np.empty((max(n, 1), 3)); oc = np.empty((max(n, 1), 3))
        m = C.c_int64(0)
        self._chk(self.L.visma_icp_voxel_down_sample(
            self._h, _p(p, _dp), n, None if nn is None else _p(nn, _dp), None if cc is None else _p(cc, _dp),
            float(voxel_size), _p(op, _dp), _p(on, _dp), _p(oc, _dp), C.byref(m)))
        m = m.value
        return op[:m].copy(), (None if nn is None else on[:m].copy()), (None if cc is None else oc[:m].copy())

    def sample_mesh(self, V, F, n, quirks=False, seed=0, uniforms=None):
        """feh::SamplePointCloudFromMesh on the GPU -> (m, 3) points, m <= n."""
        V = _f64(V, (-1, 3)); F = np.ascontiguousarray(F, np.int32).reshape(-1, 3)
        u = None if uniforms is None else _f64(uniforms, (-1, 3))
        if u is not None:
            n = len(u)
        out = np.empty((max(n, 1), 3)); m = C.c_int64(0)
        self._chk(self.L.visma_icp_sample_mesh(self._h, _p(V, _dp), len(V), _p(F, _ip), len(F), int(n),
                                               int(bool(quirks)), int(seed), None if u is None else _p(u, _dp),
                                               _p(out, _dp), C.byref(m)))
        return out[:m.value].copy()

    def point_mesh_distance(self, P, V, F):
        """-> (d2, face, closest) of every query point against the triangle mesh."""
        P = _f64(P, (-1, 3)); V = _f64(V, (-1, 3)); F = np.ascontiguousarray(F, np.int32).reshape(-1, 3)
        d2 = np.empty(max(len(P), 1)); face = np.empty(max(len(P), 1), np.int32); cl = np.empty((max(len(P), 1), 3))
        self._chk(self.L.visma_icp_point_mesh_distance(self._h, _p(P, _dp), len(P), _p(V, _dp), len(V),
                                                       _p(F, _ip), len(F), _p(d2, _dp), _p(face, _ip), _p(cl, _dp)))
        return d2[:len(P)], face[:len(P)], cl[:len(P)]

    def last_mesh_kernel_ms(self):
        """-> (query kernel ms, search-structure build ms) of the last mesh-distance call."""
        ms, bms = C.c_double(0), C.c_double(0)
        self._chk(self.L.visma_icp_last_mesh_kernel_ms(self._h, C.byref(ms), C.byref(bms)))
        return ms.value, bms.value

    def set_search_precision(self, mode):
        """'exact' (default; alias 'auto': fp32 ranking, near-ties re-ranked in f64) | 'f32' | 'f64';
        before set_clouds_f64."""
        self._chk(self.L.visma_icp_set_search_precision(self._h, {"f32": 0, "auto": 1, "exact": 1, "f64": 2}[mode]))

    def search_mode_used(self):
        """'f32' | 'exact' | 'f64': the arithmetic of the last pass."""
        v = C.c_int(0)
        self._chk(self.L.visma_icp_get_search_precision_used(self._h, C.byref(v)))
        return {0: "f32", 1: "exact", 2: "f64"}[v.value]

    def search_is_f64(self):
        """True when the last pass returned the reference's own (f64) correspondences."""
        return self.search_mode_used() != "f32"

    def set_mesh_search(self, method):
        """'auto' | 'brute' | 'bvh'"""
        self._chk(self.L.visma_icp_set_mesh_search(self._h, {"auto": 0, "brute": 1, "bvh": 2}[method]))

    def measure_surface_error(self, Vs, Fs, Vt, Ft, num_samples, quirks=False, seed=0):
        Vs = _f64(Vs, (-1, 3)); Fs = np.ascontiguousarray(Fs, np.int32).reshape(-1, 3)
        Vt = _f64(Vt, (-1, 3)); Ft = np.ascontiguousarray(Ft, np.int32).reshape(-1, 3)
        out = np.empty(5)
        self._chk(self.L.visma_icp_measure_surface_error(self._h, _p(Vs, _dp), len(Vs), _p(Fs, _ip), len(Fs),
                                                         _p(Vt, _dp), len(Vt), _p(Ft, _ip), len(Ft),
                                                         int(num_samples), int(bool(quirks)), int(seed), _p(out, _dp)))
        return dict(zip(("mean", "std", "median", "min", "max"), out))

    # ---- options ----
    def set_nn_mode(self, mode):
        self._chk(self.L.visma_icp_set_nn_mode(self._h, int(mode)))

    def nn_mode_used(self):
        m = C.c_int(0)
        self._chk(self.L.visma_icp_get_nn_mode_used(self._h, C.byref(m)))
        return m.value

    def search_kernel_used(self):
        """'brute' | 'serial' (lane-serial grid search) | 'warm' (warm-started wave-cooperative grid search) | 'ring' (cells
        smaller than the radius, searched in rings: grid_ring.hip)."""
        v = C.c_int(0)
        self._chk(self.L.visma_icp_get_search_kernel_used(self._h, C.byref(v)))
        return {0: "brute", 1: "serial", 2: "warm", 3: "ring"}[v.value]

    def forget_winners(self):
        """The next pass runs like the first of a new registration (no warm start)."""
        self._chk(self.L.visma_icp_forget_winners(self._h))

    def set_persistent(self, on=True, timeout_ms=0.0):
        """the persistent launch of a host loop (default on); timeout_ms > 0: the launch's patience"""
        self._chk(self.L.visma_icp_set_persistent(self._h, int(bool(on)), float(timeout_ms)))

    def persistent_info(self):
        """what the persistent launches of this context's host loops did so far (visma_icp_get_persistent_info)"""
        t = CPersistentInfo()
        t.struct_size = C.sizeof(CPersistentInfo)
        self._chk(self.L.visma_icp_get_persistent_info(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in CPersistentInfo._fields_ if k != "reserved"}

    def sweep_info(self):
        """persistent sweep launches of this context's device-resident loops: started / gave up (visma_icp_get_sweep_info)"""
        a, b = C.c_double(0.0), C.c_double(0.0)
        self._chk(self.L.visma_icp_get_sweep_info(self._h, C.byref(a), C.byref(b)))
        return {"launches": a.value, "aborts": b.value}

    def set_ring_search(self, mode=-1):
        """cells smaller than the radius, searched in rings (grid_ring.hip): -1 by occupancy, 0 never, 1 whenever possible"""
        self._chk(self.L.visma_icp_set_ring_search(self._h, int(mode)))

    def ring_search(self):
        """what the current grid is (visma_icp_get_ring_search): rings > 0 = the ring search"""
        r, c, o = C.c_int(0), C.c_double(0.0), C.c_double(0.0)
        self._chk(self.L.visma_icp_get_ring_search(self._h, C.byref(r), C.byref(c), C.byref(o)))
        return {"rings": r.value, "cell": c.value, "occupancy": o.value}

    def test_stall_command(self, nth, ms):
        self._chk(self.L.visma_icp_test_stall_command(self._h, int(nth), float(ms)))

    def set_device_loop(self, on=True):
        """True: on-device loop, False: host loop, None: automatic (default)."""
        self._chk(self.L.visma_icp_set_device_loop(self._h, -1 if on is None else int(bool(on))))

    def set_profiling(self, on=True):
        """False/0 off, True/1 every launch, n > 1 every n-th ICP pass."""
        self._chk(self.L.visma_icp_set_profiling(self._h, int(on)))

    def get_timing(self, reset=False):
        t = CTiming()
        self._chk(self.L.visma_icp_get_timing(self._h, C.byref(t), int(bool(reset))))
        d = {k: getattr(t, k) for k, _ in CTiming._fields_}
        d["tile_phase_cycles"] = [float(x) for x in d["tile_phase_cycles"]]
        return d

    def launch_config(self):
        a = C.c_int(); b = C.c_int()
        self._chk(self.L.visma_icp_get_launch_config(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- multi-GPU ----
    def comm_ipc_export(self):
        """-> 64-byte handle of this context's all-reduce mailbox (all-gather them, then comm_ipc_init)."""
        buf = C.create_string_buffer(IPC_HANDLE_BYTES)
        self._chk(self.L.visma_icp_comm_ipc_export(self._h, buf))
        return buf.raw

    def comm_ipc_init(self, rank, nranks, handles):
        """handles: the nranks exported handles in rank order (bytes, nranks * 64)."""
        raw = b"".join(bytes(h) for h in handles) if not isinstance(handles, (bytes, bytearray)) else bytes(handles)
        assert len(raw) == nranks * IPC_HANDLE_BYTES
        buf = C.create_string_buffer(raw, len(raw))
        self._chk(self.L.visma_icp_comm_ipc_init(self._h, int(rank), int(nranks), buf))

    def comm_init(self, rank, nranks, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        self._chk(self.L.visma_icp_comm_init(self._h, int(rank), int(nranks), buf))

    def set_allreduce(self, fn, rank, nranks):
        """fn(np.ndarray[38]) must sum in place across ranks (host all-reduce)."""
        def tramp(_user, ptr, n):
            try:
                a = np.ctypeslib.as_array(ptr, shape=(n,))
                fn(a)
                return 0
            except Exception:
                return 1
        cb = ALLREDUCE_FN(tramp)
        self._keep.append(cb)
        self._chk(self.L.visma_icp_set_allreduce(self._h, cb, None, int(rank), int(nranks)))

    def set_target_shard(self, global_offset, global_nt, centre=None):
        """Target-sharded rank: this context holds targets [offset, offset + nt) of `global_nt`.
        Call before set_clouds_f64; `centre` must be the same on every rank."""
        c = None if centre is None else _f64(centre, (3,))
        self._chk(self.L.visma_icp_set_target_shard(self._h, int(global_offset), int(global_nt),
                                                    None if c is None else _p(c, _dp)))

    def set_minreduce(self, fn):
        """fn(np.ndarray[uint64, n]) must take the element-wise minimum across ranks in place."""
        def tramp(_user, ptr, n):
            try:
                fn(np.ctypeslib.as_array(ptr, shape=(n,)))
                return 0
            except Exception:
                return 1
        cb = MINREDUCE_FN(tramp)
        self._keep.append(cb)
        self._chk(self.L.visma_icp_set_minreduce(self._h, cb, None))

    def set_global_source_count(self, n):
        self._chk(self.L.visma_icp_set_global_source_count(self._h, int(n)))


class Corpus:
    """visma_icp_run_corpus: items = [(model_xyz, scene_xyz), ...] (arrays kept alive here), one host thread per
    context pulls chunks from a counter.  `counter_address`: address of an int64 in shared memory when ranks in
    other processes pull from the same queue (else a private counter)."""

    def __init__(self, items, level=24, max_dist=0.05, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6,
                 solver=SOLVER_KABSCH, chunk=8):
        self.L = load()
        self._keep = [(_f64(m, (-1, 3)), _f64(s, (-1, 3))) for m, s in items]
        self.n = len(self._keep)
        self.items = (CCorpusItem * max(self.n, 1))()
        for i, (m, s) in enumerate(self._keep):
            self.items[i] = CCorpusItem(_p(m, _dp), len(m), _p(s, _dp), len(s))
        self.params = CCorpusParams(int(level), float(max_dist), int(max_iter), float(rel_fitness), float(rel_rmse),
                                    int(solver), int(chunk))
        self.results = (CCorpusResult * max(self.n, 1))()

    def run(self, ctxs, counter_address=None):
        """-> list of (Result best, best_level, device, iterations_all_starts) per item (device -1: another process's)."""
        arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
        err = C.create_string_buffer(512)
        cnt = C.cast(C.c_void_p(counter_address), C.POINTER(C.c_int64)) if counter_address else None
        rc = self.L.visma_icp_run_corpus(arr, len(ctxs), self.items, self.n, C.byref(self.params), cnt, self.results, err, 512)
        if rc != 0:
            raise IcpError(rc, err.value.decode(errors="replace"))
        return [(Result(self.results[i].best), int(self.results[i].best_level), int(self.results[i].device),
                 int(self.results[i].iterations_all_starts)) for i in range(self.n)]


def run_batch_multi(ctxs, problems, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6, solver=SOLVER_KABSCH):
    """visma_icp_run_batch_multi: one batch over several worker contexts (same GPU or several).  problems: the value
    of Context.make_batch() (or a list of (src, tgt, init, radius) tuples) -> list of Result."""
    L = load()
    arr, n, _keep, out = problems if (isinstance(problems, tuple) and len(problems) == 4 and
                                      isinstance(problems[1], int)) else ctxs[0].make_batch(problems)
    h = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    err = C.create_string_buffer(512)
    rc = L.visma_icp_run_batch_multi(h, len(ctxs), arr, n, int(max_iter), float(rel_fitness), float(rel_rmse), int(solver),
                                     out, err, 512)
    if rc != 0:
        raise IcpError(rc, err.value.decode(errors="replace"))
    return [Result(out[i]) for i in range(n)]


def set_persistent_cu_share(share):
    """per process: the largest part of a device's workgroup slots a persistent launch may hold (0 < share <= 1)"""
    rc = load().visma_icp_set_persistent_cu_share(float(share))
    if rc != 0:
        raise IcpError(rc, "visma_icp_set_persistent_cu_share(%r)" % (share,))


def device_count():
    """HIP devices visible to this process, asked through the library (no torch import: its bundled ROCm
    libraries, RCCL among them, must not be what a later dlopen in this process resolves to)."""
    return int(load().visma_icp_device_count())


def comm_unique_id():
    L = load()
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    rc = L.visma_icp_comm_unique_id(buf)
    if rc != OK:
        msg = L.visma_icp_last_error(None)
        raise IcpError(rc, msg.decode() if msg else "")
    return bytes(buf.raw)


def error_metric(errors):
    L = load()
    e = _f64(errors, (-1,)); out = np.empty(5)
    rc = L.visma_icp_error_metric(_p(e, _dp), len(e), _p(out, _dp))
    if rc != OK:
        raise IcpError(rc, "error_metric")
    return dict(zip(("mean", "std", "median", "min", "max"), out))


class CIoCloud(C.Structure):
    _fields_ = [("n", C.c_int64), ("xyz", _dp), ("n_normals", C.c_int64), ("normals", _dp),
                ("n_colors", C.c_int64), ("colors", _dp), ("n_faces", C.c_int64), ("faces", _ip)]


class IoError(RuntimeError):
    pass


def read_ply(path, _fn="visma_io_read_ply"):
    """open3d::ReadPointCloudFromPLY / ReadTriangleMeshFromPLY -> dict(xyz, normals, colors, faces)."""
    L = load()
    L.visma_io_last_error.restype = C.c_char_p
    c = CIoCloud()
    rc = getattr(L, _fn)(str(path).encode(), C.byref(c))
    if rc != OK:
        raise IoError("%s: %s" % (_fn, L.visma_io_last_error().decode()))
    def take(ptr, n, cols, dt):
        if n == 0:
            return np.zeros((0, cols), dt)
        return np.ctypeslib.as_array(ptr, shape=(n * cols,)).astype(dt).reshape(n, cols).copy()
    out = dict(xyz=take(c.xyz, c.n, 3, np.float64), normals=take(c.normals, c.n_normals, 3, np.float64),
               colors=take(c.colors, c.n_colors, 3, np.float64), faces=take(c.faces, c.n_faces, 3, np.int32))
    L.visma_io_free_cloud(C.byref(c))
    return out


def read_pcd(path):
    """open3d::ReadPointCloudFromPCD -> dict(xyz, normals, colors, faces=empty)."""
    return read_ply(path, "visma_io_read_pcd")


class CIoPose(C.Structure):
    _fields_ = [("name", C.c_char * 256), ("id", C.c_int), ("status", C.c_int), ("T", C.c_double * 12)]


def _poses(fn, *args):
    L = load()
    L.visma_io_last_error.restype = C.c_char_p
    L.visma_io_free.argtypes = [C.c_void_p]
    ptr = C.POINTER(CIoPose)()
    n = C.c_int64(0)
    rc = getattr(L, fn)(*args, C.byref(ptr), C.byref(n))
    if rc != OK:
        raise IoError("%s: %s" % (fn, L.visma_io_last_error().decode()))
    out = [dict(name=ptr[i].name.decode(), id=ptr[i].id, status=ptr[i].status,
                T=np.array(list(ptr[i].T)).reshape(3, 4)) for i in range(n.value)]
    L.visma_io_free(ptr)
    return out


# ---- the gravity alignment and pose composition of feh::AnnotationTool (host arithmetic, no context) -------------
def find_plane_normal(xyz):
    """feh::FindPlaneNormal (include/geometry.h:18-26), with Eigen's sign of the singular vector."""
    xyz = _f64(xyz, (-1, 3)); out = np.empty(3)
    rc = load().visma_geom_find_plane_normal(_p(xyz, _dp), C.c_int64(len(xyz)), _p(out, _dp))
    if rc != OK:
        raise IcpError(rc, "visma_geom_find_plane_normal")
    return out


def jacobi_svd3(A):
    """Eigen::JacobiSVD<Matrix3d>(A, ComputeFullU | ComputeFullV) -> (U, S, V)."""
    A = _f64(A, (9,)); U = np.empty(9); S = np.empty(3); V = np.empty(9)
    rc = load().visma_geom_jacobi_svd3(_p(A, _dp), _p(U, _dp), _p(S, _dp), _p(V, _dp))
    if rc != OK:
        raise IcpError(rc, "visma_geom_jacobi_svd3")
    return U.reshape(3, 3), S, V.reshape(3, 3)


def rotation_between_vectors(u, v):
    """feh::RotationBetweenVectors (core/utils.h:229-233)."""
    u = _f64(u, (3,)); v = _f64(v, (3,)); R = np.empty(9)
    rc = load().visma_geom_rotation_between_vectors(_p(u, _dp), _p(v, _dp), _p(R, _dp))
    if rc != OK:
        raise IcpError(rc, "visma_geom_rotation_between_vectors")
    return R.reshape(3, 3)


def centre_on_floor(xyz):
    """(-mean_x, -min_y, -mean_z): the translation of T1 / T2 (src/annotation.cpp:114-119, 128-132)."""
    xyz = _f64(xyz, (-1, 3)); t = np.empty(3)
    rc = load().visma_geom_centre_on_floor(_p(xyz, _dp), C.c_int64(len(xyz)), _p(t, _dp))
    if rc != OK:
        raise IcpError(rc, "visma_geom_centre_on_floor")
    return t


def annot_total_pose(T0, T1, T2, T3):
    """Ttot = (T1 T0)^-1 T3 T2 (src/annotation.cpp:147-153)."""
    a = [_f64(T, (16,)) for T in (T0, T1, T2, T3)]; out = np.empty(16)
    rc = load().visma_annot_total_pose(*[_p(x, _dp) for x in a], _p(out, _dp))
    if rc != OK:
        raise IcpError(rc, "visma_annot_total_pose")
    return out.reshape(4, 4)


def read_alignment_json(path):
    """alignment.json (name -> 3x4 pose) -> list of dict(name, T), in key order like the reference's loop."""
    return _poses("visma_io_read_alignment_json", str(path).encode())


def read_result_json(path, packet=-1):
    """result.json -> the objects of one packet (default: the last): dict(id, status, name, T)."""
    return _poses("visma_io_read_result_json", str(path).encode(), C.c_int64(packet))


def write_alignment_json(path, poses):
    """poses: iterable of (name, 3x4 or 4x4 matrix)."""
    L = load()
    L.visma_io_last_error.restype = C.c_char_p
    poses = list(poses)
    arr = (CIoPose * max(len(poses), 1))()
    for i, (name, T) in enumerate(poses):
        arr[i].name = str(name).encode()
        T = np.asarray(T, np.float64)[:3, :4]
        arr[i].T = (C.c_double * 12)(*T.reshape(12))
    rc = L.visma_io_write_alignment_json(str(path).encode(), arr, C.c_int64(len(poses)))
    if rc != OK:
        raise IoError("visma_io_write_alignment_json: %s" % L.visma_io_last_error().decode())


def read_obj(path):
    """igl::readOBJ(path, V, F) -> (V [nv x 3], F [nf x face_size])."""
    L = load()
    L.visma_io_last_error.restype = C.c_char_p
    L.visma_io_free.argtypes = [C.c_void_p]
    V, F = _dp(), _ip()
    nv, nf, fs = C.c_int64(0), C.c_int64(0), C.c_int(0)
    rc = L.visma_io_read_obj(str(path).encode(), C.byref(V), C.byref(nv), C.byref(F), C.byref(nf), C.byref(fs))
    if rc != OK:
        raise IoError("visma_io_read_obj: %s" % L.visma_io_last_error().decode())
    v = np.ctypeslib.as_array(V, shape=(max(3 * nv.value, 1),))[:3 * nv.value].reshape(-1, 3).copy()
    f = np.ctypeslib.as_array(F, shape=(max(nf.value * fs.value, 1),))[:nf.value * fs.value].reshape(nf.value, max(fs.value, 0)).copy()
    L.visma_io_free(V); L.visma_io_free(F)
    return v, f


def tile_config():
    L = load()
    a = C.c_int(); b = C.c_int(); c = C.c_int()
    L.visma_icp_get_tile_config(C.byref(a), C.byref(b), C.byref(c))
    return {"s_tile": a.value, "t_chunk": b.value, "block": c.value}


def solve_from_stats(stats, solver=SOLVER_KABSCH, with_scaling=False):
    L = load()
    st = _f64(stats, (NSTATS,)); T = np.empty(16)
    rc = L.visma_icp_solve_from_stats(_p(st, _dp), int(solver), int(bool(with_scaling)), _p(T, _dp))
    if rc != OK:
        raise IcpError(rc, "solve_from_stats")
    return T.reshape(4, 4)
