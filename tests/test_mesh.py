"""The mesh steps either side of ICP.

CPU: the oracle restatement against igl::AABB outputs (golden/mesh.npz) and
against the source-level properties of feh::SamplePointCloudFromMesh
(include/geometry.h:29-64).  GPU: the product kernels against the oracle on the
same uniforms -- same faces picked, same points to the last bit.
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def mesh():
    return dict(np.load(os.path.join(HERE, "golden", "mesh.npz")))


def _on_triangle(p, a, b, c, tol=1e-9):
    n = np.cross(b - a, c - a)
    nn = np.linalg.norm(n, axis=1)
    ok = nn > 0
    dist = np.abs(np.einsum("ij,ij->i", p - a, n)) / np.where(ok, nn, 1)
    # barycentric
    v0, v1, v2 = b - a, c - a, p - a
    d00 = np.einsum("ij,ij->i", v0, v0); d01 = np.einsum("ij,ij->i", v0, v1)
    d11 = np.einsum("ij,ij->i", v1, v1); d20 = np.einsum("ij,ij->i", v2, v0)
    d21 = np.einsum("ij,ij->i", v2, v1)
    den = np.where(ok, d00 * d11 - d01 * d01, 1)
    v = (d11 * d20 - d01 * d21) / den; w = (d00 * d21 - d01 * d20) / den
    return (dist < tol) & (v > -1e-9) & (w > -1e-9) & (v + w < 1 + 1e-9)


# ---------------------------------------------------------------------------
# CPU: oracle vs golden
# ---------------------------------------------------------------------------
def test_oracle_point_mesh_matches_igl(oracle, mesh):
    d2, face, cl = oracle.point_mesh_sqdist(mesh["P"], mesh["V"], mesh["F"])
    scale = np.maximum(mesh["igl_d2"], 1e-12)
    assert np.abs(d2 - mesh["igl_d2"]).max() < 1e-15 or (np.abs(d2 - mesh["igl_d2"]) / scale).max() < 1e-12
    # the closest POINT is unique wherever the distance is (faces can tie on shared edges)
    assert np.abs(cl - mesh["igl_closest"]).max() < 1e-9
    V, F = mesh["V"], mesh["F"]
    assert _on_triangle(cl, V[F[face, 0]], V[F[face, 1]], V[F[face, 2]], 1e-9).all()
    assert np.allclose(((mesh["P"] - cl) ** 2).sum(1), d2, rtol=1e-12, atol=1e-30)


def test_oracle_sample_mesh_reference_quirks(oracle, mesh):
    V, F, u = mesh["V"], mesh["F"], mesh["uniforms"]
    a, b, c = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    total = 0.0
    for x in area:
        total += x                                   # geometry.h:36-39, sequential
    cdf = np.empty(len(F)); cdf[0] = area[0] / total
    for i in range(1, len(F)):
        cdf[i] = cdf[i - 1] + area[i] / total        # geometry.h:40-43
    # reference mapping (geometry.h:52-57): k with cdf[k] <= r < cdf[k+1], k <= nf-2
    k = np.searchsorted(cdf, u[:, 0], side="right") - 1
    keep = (k >= 0) & (k <= len(F) - 2)
    kk = k[keep]
    exp = a[kk] + u[keep, 1:2] * (b[kk] - a[kk]) + u[keep, 2:3] * (c[kk] - a[kk])
    got = oracle.sample_mesh(V, F, u, quirks=True)
    assert got.shape == exp.shape
    assert np.abs(got - exp).max() < 1e-15
    # the documented quirk: about half of those points are off the triangle
    off = ~_on_triangle(got, a[kk], b[kk], c[kk])
    assert 0.4 < off.mean() < 0.6
    # r below cdf[0] yields NO point
    u0 = np.array([[cdf[0] * 0.5, 0.2, 0.2], [cdf[0] * 1.5, 0.2, 0.2]])
    assert len(oracle.sample_mesh(V, F, u0, quirks=True)) == 1


def test_oracle_sample_mesh_corrected(oracle, mesh):
    V, F, u = mesh["V"], mesh["F"], mesh["uniforms"]
    got = oracle.sample_mesh(V, F, u, quirks=False)
    assert len(got) == len(u)
    d2, _, _ = oracle.point_mesh_sqdist(got, V, F)
    assert d2.max() < 1e-24                          # every sample lies on the surface


def test_oracle_error_metric(oracle):
    rng = np.random.default_rng(5)
    e = rng.random(1001)
    m = oracle.error_metric(e)
    assert m["mean"] == pytest.approx(e.mean(), rel=1e-13)
    assert m["std"] == pytest.approx(e.std(), rel=1e-10)
    assert m["median"] == np.sort(e)[len(e) >> 1]    # geometry.h:97: sorted[n >> 1]
    assert m["min"] == e.min() and m["max"] == e.max()
    e2 = rng.random(10)
    assert oracle.error_metric(e2)["median"] == np.sort(e2)[5]   # upper median for even n


def test_error_metric_host(lib, oracle):
    from visma_amd import _lib
    e = np.random.default_rng(6).random(777)
    got, exp = _lib.error_metric(e), oracle.error_metric(e)
    assert got == exp


# ---------------------------------------------------------------------------
# GPU: product vs oracle / golden
# ---------------------------------------------------------------------------
@pytest.fixture(params=["brute", "bvh"])
def mesh_ctx(request, gpu_ctx_auto):
    gpu_ctx_auto.set_mesh_search(request.param)
    yield gpu_ctx_auto
    gpu_ctx_auto.set_mesh_search("auto")


@pytest.mark.gpu
def test_gpu_point_mesh_distance(mesh_ctx, oracle, mesh):
    gpu_ctx_auto = mesh_ctx
    d2, face, cl = gpu_ctx_auto.point_mesh_distance(mesh["P"], mesh["V"], mesh["F"])
    od2, oface, ocl = oracle.point_mesh_sqdist(mesh["P"], mesh["V"], mesh["F"])
    assert np.array_equal(d2, od2)                   # same arithmetic, no contraction: bit-exact
    assert np.array_equal(face, oface)               # lowest face index on exact ties
    assert np.array_equal(cl, ocl)
    assert np.abs(cl - mesh["igl_closest"]).max() < 1e-9
    assert np.allclose(d2, mesh["igl_d2"], rtol=1e-12, atol=1e-15)


@pytest.mark.gpu
def test_gpu_bvh_equals_brute_force_on_a_scene(gpu_ctx_auto, mesh):
    """10 chairs, queries near, on and far from the surface, duplicated faces (exact ties)."""
    V, F = mesh["V"], mesh["F"]
    rng = np.random.default_rng(9)
    Vs, Fs = [], []
    for i in range(10):
        c, s_ = np.cos(0.7 * i), np.sin(0.7 * i)
        R = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]])
        Vs.append(V @ R.T + [2.0 * (i % 4), 0.0, 2.0 * (i // 4)]); Fs.append(F + i * len(V))
    Vs = np.concatenate(Vs); Fs = np.concatenate(Fs + [Fs[3][:500]]).astype(np.int32)   # 500 duplicated faces
    P = np.concatenate([gpu_ctx_auto.sample_mesh(Vs, Fs, 40000, seed=5) + rng.standard_normal((40000, 3)) * 0.02,
                        gpu_ctx_auto.sample_mesh(Vs, Fs, 5000, seed=6),
                        rng.uniform(-3, 9, (5000, 3)), Vs[::50]])
    gpu_ctx_auto.set_mesh_search("brute")
    a = gpu_ctx_auto.point_mesh_distance(P, Vs, Fs)
    gpu_ctx_auto.set_mesh_search("bvh")
    b = gpu_ctx_auto.point_mesh_distance(P, Vs, Fs)
    gpu_ctx_auto.set_mesh_search("auto")
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.gpu
def test_gpu_point_mesh_edge_cases(mesh_ctx, oracle, mesh):
    gpu_ctx_auto = mesh_ctx
    V, F = mesh["V"], mesh["F"]
    # one triangle, one point per Voronoi region
    tv = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.0]]); tf = np.array([[0, 1, 2]], np.int32)
    P = np.array([[-1, -1, 0.5], [2, -1, 0], [-1, 2, 0], [0.5, -1, 0], [-1, 0.5, 0], [1, 1, 0], [0.2, 0.2, 3.0]])
    d2, face, cl = gpu_ctx_auto.point_mesh_distance(P, tv, tf)
    od2, _, ocl = oracle.point_mesh_sqdist(P, tv, tf)
    assert np.array_equal(d2, od2) and np.array_equal(cl, ocl) and (face == 0).all()
    assert d2[-1] == 9.0 and d2[0] == 2.25
    # degenerate (zero-area) faces and a face count that is not a multiple of the LDS tile
    Fd = np.concatenate([F[:131], [[5, 5, 5], [7, 8, 7]]]).astype(np.int32)
    d2, face, cl = gpu_ctx_auto.point_mesh_distance(mesh["P"][:257], V, Fd)
    od2, oface, ocl = oracle.point_mesh_sqdist(mesh["P"][:257], V, Fd)
    assert np.array_equal(d2, od2) and np.array_equal(face, oface)
    # empty query set; out-of-range face index is refused
    d2, _, _ = gpu_ctx_auto.point_mesh_distance(np.zeros((0, 3)), V, F)
    assert len(d2) == 0
    from visma_amd._lib import IcpError
    with pytest.raises(IcpError):
        gpu_ctx_auto.point_mesh_distance(P, tv, np.array([[0, 1, 3]], np.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("quirks", [True, False])
def test_gpu_sample_mesh_matches_oracle(gpu_ctx_auto, oracle, mesh, quirks):
    V, F, u = mesh["V"], mesh["F"], mesh["uniforms"].copy()
    a = V[F[0, 0]]; b = V[F[0, 1]]; c = V[F[0, 2]]
    area0 = 0.5 * np.linalg.norm(np.cross(b - a, c - a))
    u[3, 0] = 0.0                                    # r < cdf[0]: dropped by the reference mapping
    u[9, 0] = 1.0 - 2 ** -53                         # the very end of the table
    got = gpu_ctx_auto.sample_mesh(V, F, 0, quirks=quirks, uniforms=u)
    exp = oracle.sample_mesh(V, F, u, quirks=quirks)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp)
    assert len(got) == (len(u) - 1 if quirks else len(u)) or area0 == 0


@pytest.mark.gpu
def test_gpu_sample_mesh_philox(gpu_ctx_auto, oracle, mesh):
    V, F = mesh["V"], mesh["F"]
    n = 200000
    p1 = gpu_ctx_auto.sample_mesh(V, F, n, quirks=False, seed=42)
    p2 = gpu_ctx_auto.sample_mesh(V, F, n, quirks=False, seed=42)
    p3 = gpu_ctx_auto.sample_mesh(V, F, n, quirks=False, seed=43)
    assert p1.shape == (n, 3) and np.array_equal(p1, p2) and not np.array_equal(p1, p3)
    # on the surface
    d2, face, _ = gpu_ctx_auto.point_mesh_distance(p1[:20000], V, F)
    assert d2.max() < 1e-24
    # area-uniform: face histogram against the areas (chi-square, 4999 bins, 200k draws)
    a, b, c = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    big = np.argsort(-area)[:50]
    d2, face, _ = gpu_ctx_auto.point_mesh_distance(p1[:50000], V, F)
    frac = np.array([(face == f).mean() for f in big])
    expf = area[big] / area.sum()
    # ties go to the lowest face, so coplanar neighbours can steal a little: loose bound
    assert np.abs(frac.sum() - expf.sum()) < 0.02
    # the reference mapping loses points with probability cdf[0]
    q = gpu_ctx_auto.sample_mesh(V, F, n, quirks=True, seed=42)
    assert n - 50 <= len(q) <= n


@pytest.mark.gpu
def test_gpu_measure_surface_error(gpu_ctx_auto, oracle, mesh):
    V, F, Vt = mesh["V"], mesh["F"], mesh["Vt"]
    # explicit pipeline on the oracle with the same Philox points
    pts = gpu_ctx_auto.sample_mesh(V, F, 5000, quirks=False, seed=7)
    od2, _, _ = oracle.point_mesh_sqdist(pts, Vt, F)
    exp = oracle.error_metric(np.sqrt(od2))
    got = gpu_ctx_auto.measure_surface_error(V, F, Vt, F, 5000, quirks=False, seed=7)
    for k in exp:
        assert got[k] == exp[k], k
    assert 0 < got["mean"] < 0.02
    # a mesh against itself: zero error
    z = gpu_ctx_auto.measure_surface_error(V, F, V, F, 2000, quirks=False, seed=1)
    assert z["max"] < 1e-12
