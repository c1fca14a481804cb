#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/gen_golden.py

Every expected value below is an OUTPUT of the reference's own compiled code
(oracle/_ref/libvisma_ref.so = Open3D 0.3.0 RegistrationICP / estimators /
PointCloud / Eigen utilities + VISMA core/rodrigues.h), or a literal the
reference's unit tests hold (the two Open3D known-answer tests).  Inputs are
stored float32-rounded so every implementation sees bit-identical points.
Only data is written: no reference source travels.
"""
import ctypes
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.oracle import Ref, EST_POINT_TO_PLANE, EST_POINT_TO_POINT  # noqa: E402

REF = "/root/reference"
O3D_DATA = REF + "/thirdparty/Open3D/examples/TestData"


def f32(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32)


def load_obj(path):
    V, F = [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            V.append([float(x) for x in p[1:4]])
        elif p[0] == "f":
            F.append([int(x.split("/")[0]) - 1 for x in p[1:4]])
    return np.array(V), np.array(F)


def sample_mesh(V, F, n, seed):
    """Seeded area-uniform surface sampling (our own sampler: the reference's
    include/geometry.h:29-64 is time-seeded, so both sides are fed these points)."""
    rng = np.random.Generator(np.random.Philox(seed))
    a, b, c = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    tri = rng.choice(len(F), size=n, p=area / area.sum())
    u, v = rng.random(n), rng.random(n)
    flip = u + v > 1
    u[flip], v[flip] = 1 - u[flip], 1 - v[flip]
    return a[tri] + u[:, None] * (b[tri] - a[tri]) + v[:, None] * (c[tri] - a[tri])


def load_pcd_xyz_normals(path):
    raw = open(path, "rb").read()
    head_end = raw.index(b"DATA binary\n") + len(b"DATA binary\n")
    header = raw[:head_end].decode("ascii", "replace")
    n = int([l for l in header.splitlines() if l.startswith("POINTS")][0].split()[1])
    fields = [l for l in header.splitlines() if l.startswith("FIELDS")][0].split()[1:]
    arr = np.frombuffer(raw[head_end:head_end + n * 4 * len(fields)], dtype=np.float32)
    arr = arr.reshape(n, len(fields))
    xyz = arr[:, :3].astype(np.float64)
    nrm = arr[:, 3:6].astype(np.float64)
    ok = np.isfinite(xyz).all(1) & np.isfinite(nrm).all(1)
    return xyz[ok], nrm[ok]


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def make_T(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def trace(ref, src, tgt, r, init, iters, **kw):
    """Per-iteration (T, fitness, rmse, K): RegistrationICP with max_iter = 0..iters."""
    rows = []
    last = None
    for m in range(iters + 1):
        last = ref.registration_icp(src, tgt, r, init=init, max_iter=m, rel_fitness=0.0,
                                    rel_rmse=0.0, **kw)
        rows.append(np.concatenate([last.T.ravel(), [last.fitness, last.rmse, last.k]]))
    return np.array(rows), last


def glibc_rand_points(n, lo=0.0, hi=1000.0):
    """UnitTest::Rand (O3D/UnitTest/UnitTest.cpp:76-95): srand(0); x,y,z = rand()."""
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(0)
    RAND_MAX = 2147483647
    f = (hi - lo) / RAND_MAX
    out = np.empty((n, 3))
    for i in range(n):
        for a in range(3):
            out[i, a] = lo + libc.rand() * f
    return out


def gen_voxel(ref):
    """VoxelDownSample outputs of the reference, rows sorted lexicographically
    (the reference's own order is its hash map's iteration order)."""
    a_xyz, a_n = load_pcd_xyz_normals(O3D_DATA + "/Feature/cloud_bin_0.pcd")
    a_xyz = f32(a_xyz).astype(np.float64); a_n = f32(a_n).astype(np.float64)
    rng = np.random.default_rng(61)
    col = rng.random(a_xyz.shape)
    a_n[7] = np.nan                       # a NaN normal must be skipped (DownSample.cpp:53-58)
    out = dict(xyz=a_xyz, normals=a_n, colors=col)
    for name, v in (("v005", 0.05), ("v02", 0.2), ("v1em4", 1e-4)):
        p, n, c = ref.voxel_down_sample(a_xyz, v, a_n, col)
        k = np.lexsort((p[:, 2], p[:, 1], p[:, 0]))
        out[name + "_size"] = v
        out[name + "_p"], out[name + "_n"], out[name + "_c"] = p[k], n[k], c[k]
    # O3D/UnitTest/Core/Geometry/PointCloud.cpp:677-783: 20 random points, voxel 0.5
    pts = glibc_rand_points(20)
    out["ka_points"] = pts
    out["ka_ref_first"] = np.array([352.458347, 807.724520, 919.026474])   # literal ref_points[0]
    p, _, _ = ref.voxel_down_sample(pts, 0.5)
    # the literal is one of the output voxels (each point sits alone in its voxel);
    # WHERE it appears depends on the hash map of the C++ library in use
    assert np.abs(p - out["ka_ref_first"]).max(1).min() < 1e-6
    out["ka_out_sorted"] = p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]
    np.savez_compressed(os.path.join(HERE, "voxel.npz"), **out)
    print("voxel fixture:", {k: v.shape for k, v in out.items() if k.endswith("_p")})


def gen_mesh(ref):
    """Point -> mesh distances of igl::AABB (thirdparty/libigl, as
    include/geometry.h:123-136 calls it) on the reference's chair mesh."""
    V, F = load_obj(REF + "/misc/hermanmiller_aeron.obj")
    F = F.astype(np.int32)
    rng = np.random.Generator(np.random.Philox(71))
    near = sample_mesh(V, F, 1500, 72) + rng.standard_normal((1500, 3)) * 0.01
    on = sample_mesh(V, F, 300, 73)                      # on the surface (d ~ 0)
    far = rng.uniform(-1.5, 1.5, (400, 3))
    verts = V[rng.choice(len(V), 100, replace=False)]    # exact vertices: ties between faces
    mids = 0.5 * (V[F[:100, 0]] + V[F[:100, 1]])         # on edges
    P = np.concatenate([near, on, far, verts, mids])
    d2, _, cl = ref.point_mesh_sqdist(P, V, F)
    # MeasureSurfaceError inputs: the chair against a slightly moved copy of itself
    T = make_T(rot_y(math.radians(2.0)), [0.004, -0.002, 0.003])
    Vt = V @ T[:3, :3].T + T[:3, 3]
    u = rng.random((2000, 3))
    np.savez_compressed(os.path.join(HERE, "mesh.npz"), V=V, F=F, P=P, igl_d2=d2, igl_closest=cl,
                        Vt=Vt, uniforms=u)
    print("mesh fixture:", V.shape, F.shape, P.shape, "max d", math.sqrt(d2.max()))


def _ply_header(fmt, elems):
    lines = ["ply", "format %s 1.0" % fmt, "comment written by tests/golden/gen_golden.py", "obj_info a b c"]
    for name, count, props in elems:
        lines.append("element %s %d" % (name, count))
        lines += ["property " + p for p in props]
    lines.append("end_header")
    return ("\n".join(lines) + "\n").encode()


def gen_io(ref):
    """On-disk formats: small files in the layouts VISMA's callers read, and what
    the reference's readers (rply-based ReadPointCloudFromPLY / ReadTriangleMeshFromPLY,
    igl::readOBJ) return for them.  The files are written by THIS script (our own writer)
    except cube.ply, a 640-byte data file of the reference (misc/cube.ply)."""
    import shutil, struct
    d = os.path.join(HERE, "io")
    os.makedirs(d, exist_ok=True)
    shutil.copyfile(REF + "/misc/cube.ply", os.path.join(d, "cube.ply"))
    rng = np.random.default_rng(91)
    n = 300
    xyz = rng.standard_normal((n, 3))
    nrm = rng.standard_normal((n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    col = rng.integers(0, 256, (n, 3))
    qual = rng.random(n); lab = rng.integers(-5, 5, n)
    faces = [list(rng.integers(0, n, 3)) for _ in range(120)] + [list(rng.integers(0, n, 4)) for _ in range(5)] \
        + [list(rng.integers(0, n, 2))]                       # triangles, quads, one 2-gon
    # (1) binary little endian: float xyz + extra properties in between + normals + uchar colours; faces uchar/int
    b = _ply_header("binary_little_endian", [
        ("vertex", n, ["float x", "float y", "float quality", "float z", "float nx", "float ny", "float nz",
                       "int label", "uchar red", "uchar green", "uchar blue"]),
        ("face", len(faces), ["list uchar int vertex_indices"])])
    for i in range(n):
        b += struct.pack("<fffffffiBBB", xyz[i, 0], xyz[i, 1], qual[i], xyz[i, 2], *nrm[i], lab[i], *col[i])
    for f in faces:
        b += struct.pack("<B%di" % len(f), len(f), *f)
    open(os.path.join(d, "gen_le.ply"), "wb").write(b)
    # (2) binary big endian: double xyz, ushort colours, short normals-like ints; faces "vertex_index" uint8/uint
    b = _ply_header("binary_big_endian", [
        ("vertex", n, ["double x", "double y", "double z", "ushort red", "ushort green", "ushort blue",
                       "short nx", "short ny", "short nz"]),
        ("face", 100, ["list uint8 uint32 vertex_index"])])
    for i in range(n):
        b += struct.pack(">dddHHHhhh", *xyz[i], *(col[i] * 200), *(nrm[i] * 1000).astype(int))
    for f in faces[:100]:
        b += struct.pack(">B%dI" % len(f), len(f), *f)
    open(os.path.join(d, "gen_be.ply"), "wb").write(b)
    # (3) ASCII: full-precision text in "float" properties, face element BEFORE the vertex element
    t = _ply_header("ascii", [("face", 60, ["list uchar int vertex_indices"]),
                              ("vertex", n, ["float x", "float y", "float z", "float nx", "float ny", "float nz",
                                             "uchar red", "uchar green", "uchar blue"])]).decode()
    for f in faces[:60]:
        t += " ".join(str(v) for v in [len(f)] + list(f)) + "\n"
    for i in range(n):
        t += "%.17g %.9g %r %.6f %.6f %.6f %d %d %d\n" % (xyz[i, 0], xyz[i, 1], float(xyz[i, 2]), *nrm[i], *col[i])
    open(os.path.join(d, "gen_ascii.ply"), "w").write(t)
    # (4) vertices only, no normals / colours
    b = _ply_header("binary_little_endian", [("vertex", 17, ["float x", "float y", "float z"])])
    b += xyz[:17].astype("<f4").tobytes()
    open(os.path.join(d, "gen_points.ply"), "wb").write(b)
    # (5) broken: truncated body / no vertex element
    open(os.path.join(d, "bad_truncated.ply"), "wb").write(b[:-20])
    open(os.path.join(d, "bad_novertex.ply"), "wb").write(_ply_header("ascii", [("face", 0, ["list uchar int vertex_indices"])]))
    # ---- OBJ
    V, F = load_obj(REF + "/misc/hermanmiller_aeron.obj")
    V, F = V[:400], F[(F < 400).all(1)][:500]
    def obj(name, body):
        open(os.path.join(d, name), "w").write(body)
    obj("tri.obj", "# comment\no thing\n" + "".join("v %.9g %.9g %.9g\n" % tuple(v) for v in V)
        + "".join("vn 0 0 1\nvt 0.5 0.5\n" for _ in range(3)) + "g grp\ns off\n"
        + "".join("f %d/1/1 %d/2/2 %d/3/3\n" % tuple(f + 1) for f in F))
    obj("mixed_syntax.obj", "".join("v %r %r %r\n" % tuple(float(x) for x in v) for v in V) + "vn 0 0 1\n"
        + "".join(("f %d %d %d\n", "f %d//1 %d//1 %d//1\n", "f %d/1 %d/1 %d/1\n")[i % 3] % tuple(f + 1)
                  for i, f in enumerate(F)) + "vt 0 0\n")
    body = ""
    for i, f in enumerate(F[:200]):                      # negative (relative) indices, vertices interleaved
        body += "".join("v %.9g %.9g %.9g\n" % tuple(V[j]) for j in f) + "f -3 -2 -1\n"
    obj("negative.obj", body)
    quads = np.concatenate([F[:50], F[50:100, :1]], axis=1)
    obj("quads.obj", "".join("v %.9g %.9g %.9g 1.0\n" % tuple(v) for v in V) + "".join("f %d %d %d %d\n" % tuple(q + 1) for q in quads))
    obj("colors.obj", "".join("v %.9g %.9g %.9g 0.1 0.2 0.3\n" % tuple(v) for v in V) + "".join("f %d %d %d\n" % tuple(f + 1) for f in F[:20]))
    obj("bad_mixed_faces.obj", "".join("v %.9g %.9g %.9g\n" % tuple(v) for v in V[:10]) + "f 1 2 3\nf 1 2 3 4\n")
    obj("bad_short_vertex.obj", "v 1 2\nv 1 2 3\nf 1 2 2\n")
    obj("bad_face_token.obj", "v 1 2 3\nv 1 2 3\nv 0 0 0\nf 1 x 3\n")
    out = {}
    for name in sorted(os.listdir(d)):
        path = os.path.join(d, name)
        key = name.replace(".", "_")
        if name.endswith(".pcd") or name.endswith(".json") or name.startswith("bad_hugecount") or name == "two_faces.ply":
            continue                                  # gen_pcd / hand-written hardening cases
        if name.endswith(".ply"):
            c = ref.read_ply_cloud(path)
            m = ref.read_ply_mesh(path)
            out[key + "_ok"] = np.array([c is not None, m is not None])
            if c is not None:
                for k in ("xyz", "normals", "colors"):
                    out[key + "_" + k] = c[k]
            if m is not None:
                out[key + "_faces"] = m["faces"]
        else:
            r = ref.read_obj(path)
            out[key + "_ok"] = np.array([r is not None])
            if r is not None:
                out[key + "_V"], out[key + "_F"] = r
    np.savez_compressed(os.path.join(HERE, "io.npz"), **out)
    print("io fixture:", {k: v.tolist() for k, v in out.items() if k.endswith("_ok")})


def gen_pcd(ref):
    """PCD files in the layouts PCL / Open3D write, and what the reference's ReadPointCloudFromPCD
    (O3D/IO/FileFormat/FilePCD.cpp:727-760, compiled into oracle/_ref) returns for them.  The ascii and
    binary files are written by THIS script; the binary_compressed ones by the reference's own writer."""
    import struct
    d = os.path.join(HERE, "io")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(92)
    n = 257
    xyz = rng.standard_normal((n, 3)).astype(np.float32)
    nrm = rng.standard_normal((n, 3)).astype(np.float32)
    col = rng.integers(0, 256, (n, 4)).astype(np.uint8)        # b g r a
    xyz[5, 0] = np.nan; xyz[9, 1] = np.nan; xyz[13, 2] = np.nan  # rows 5 and 9 go, row 13 stays (FilePCD.cpp:521-523)

    def hdr(fields, sizes, types, counts, width, height, data, points=None, first="FIELDS"):
        h = "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n%s %s\n" % (first, " ".join(fields))
        if sizes: h += "SIZE %s\n" % " ".join(str(x) for x in sizes)
        if types: h += "TYPE %s\n" % " ".join(types)
        if counts: h += "COUNT %s\n" % " ".join(str(x) for x in counts)
        h += "WIDTH %d\nHEIGHT %d\nVIEWPOINT 0 0 0 1 0 0 0\n" % (width, height)
        if points is not None: h += "POINTS %d\n" % points
        return (h + "DATA %s\n" % data).encode()

    rgbf = col.view(np.float32).reshape(n)                      # the 4 colour bytes read as a float (PCL's "rgb")
    # (1) ascii, normals + float-packed rgb, a short line in the data, tabs
    t = hdr(["x", "y", "z", "normal_x", "normal_y", "normal_z", "rgb"], [4] * 7, ["F"] * 7, [1] * 7, n, 1, "ascii", n).decode()
    for i in range(n):
        if i == 20: t += "1 2 3\n"                             # too few tokens: skipped (FilePCD.cpp:362-364)
        t += ("%r\t%r %r %.6g %.6g %.6g %.9g\n" % (float(xyz[i, 0]), float(xyz[i, 1]), float(xyz[i, 2]), *nrm[i], rgbf[i]))
    open(os.path.join(d, "gen_ascii.pcd"), "w").write(t)
    # (2) binary, mixed types, an extra field with COUNT 3 in between, rgba as U4, HEIGHT > 1
    b = hdr(["x", "y", "intensity", "z", "hist", "label", "rgba"], [4, 4, 4, 4, 2, 1, 4], ["F", "F", "F", "F", "U", "I", "U"],
            [1, 1, 1, 1, 3, 1, 1], 32, 8, "binary")
    for i in range(256):
        b += struct.pack("<ffff3Hb4B", xyz[i, 0], xyz[i, 1], 0.5, xyz[i, 2], 1, 2, 3, -3, *col[i])
    open(os.path.join(d, "gen_binary.pcd"), "wb").write(b)
    # (3) integer coordinates: ascii I / U with hex and octal text (strtol base 0), colour as ascii U
    t = hdr(["x", "y", "z", "rgb"], [4, 2, 1, 4], ["I", "I", "U", "U"], [1, 1, 1, 1], 6, 1, "ascii", first="COLUMNS").decode()
    t += "0x10 -7 3 255\n010 5 200 16711680\n-2147483648 32767 0 65280\n12 0x7f 9 4278190335\n1 2 3 0\n7 8 9 1\n"
    open(os.path.join(d, "gen_int.pcd"), "w").write(t)
    # (4) binary: 8-byte floats and 2-byte colour read as 0 (FilePCD.cpp:268-281, :289-296)
    b = hdr(["x", "y", "z", "rgb"], [8, 4, 4, 2], ["F", "F", "F", "U"], [1, 1, 1, 1], 9, 1, "binary")
    for i in range(9):
        b += struct.pack("<dffH", float(xyz[i, 0]), xyz[i, 1], xyz[i, 2], 7)
    open(os.path.join(d, "gen_wide.pcd"), "wb").write(b)
    # (5) the reference's own writer: binary_compressed (LZF) and ascii, with normals and colours
    keep = ~(np.isnan(xyz).any(1))
    assert ref.write_pcd(os.path.join(d, "ref_compressed.pcd"), xyz[keep].astype(np.float64), nrm[keep].astype(np.float64),
                         col[keep, :3] / 255.0, ascii=False, compressed=True)
    big = np.repeat(xyz[keep][:40].astype(np.float64), 30, axis=0) + np.arange(1200)[:, None] * 1e-3   # long LZF matches
    assert ref.write_pcd(os.path.join(d, "ref_compressed_xyz.pcd"), big, None, None, ascii=False, compressed=True)
    assert ref.write_pcd(os.path.join(d, "ref_ascii.pcd"), xyz[keep][:50].astype(np.float64), nrm[keep][:50].astype(np.float64), None, ascii=True)
    # (6) broken files
    open(os.path.join(d, "bad_truncated.pcd"), "wb").write(open(os.path.join(d, "gen_binary.pcd"), "rb").read()[:-10])
    open(os.path.join(d, "bad_nofields.pcd"), "wb").write(hdr(["a", "b", "z"], [4] * 3, ["F"] * 3, [1] * 3, 3, 1, "ascii") + b"1 2 3\n1 2 3\n1 2 3\n")
    open(os.path.join(d, "bad_nopoints.pcd"), "wb").write(hdr(["x", "y", "z"], [4] * 3, ["F"] * 3, [1] * 3, 0, 1, "ascii"))
    open(os.path.join(d, "bad_sizecount.pcd"), "wb").write(hdr(["x", "y", "z"], [4, 4], ["F"] * 3, [1] * 3, 2, 1, "ascii") + b"1 2 3\n1 2 3\n")
    comp = open(os.path.join(d, "ref_compressed.pcd"), "rb").read()
    open(os.path.join(d, "bad_lzf.pcd"), "wb").write(comp[:-7] + b"\xff" * 7)
    out = {}
    for name in sorted(os.listdir(d)):
        if not name.endswith(".pcd"):
            continue
        c = ref.read_pcd_cloud(os.path.join(d, name))
        key = name.replace(".", "_")
        out[key + "_ok"] = np.array([c is not None])
        if c is not None:
            for k in ("xyz", "normals", "colors"):
                out[key + "_" + k] = c[k]
    np.savez_compressed(os.path.join(HERE, "pcd.npz"), **out)
    print("pcd fixture:", {k: (v.tolist(), out.get(k[:-3] + "_xyz", np.zeros((0, 3))).shape) for k, v in out.items() if k.endswith("_ok")})


def gen_normals(ref):
    """open3d::EstimateNormals (O3D/Core/Geometry/EstimateNormals.cpp:114-153): outputs of the compiled
    reference for the three searches, on chair CAD samples and on a real scan fragment, with and
    without existing normals.  -> normals.npz"""
    V, F = load_obj(REF + "/misc/hermanmiller_aeron.obj")
    chair = f32(sample_mesh(V, F, 4000, 21)).astype(np.float64)
    frag, frag_n = load_pcd_xyz_normals(O3D_DATA + "/Feature/cloud_bin_0.pcd")
    frag = f32(frag[::2][:5000]).astype(np.float64)
    frag_n = f32(frag_n[::2][:5000]).astype(np.float64)
    dup = np.concatenate([chair[:300], chair[:300], chair[300:600]])      # exact duplicates: zero distances
    line = np.stack([np.linspace(0, 1, 40), np.zeros(40), np.zeros(40)], 1)   # rank-1 neighbourhoods: zero vector
    out = {"chair": f32(chair), "frag": f32(frag), "frag_normals": f32(frag_n), "dup": f32(dup), "line": f32(line)}
    cases = {
        "chair_knn30": (chair, dict(knn=30)),
        "chair_knn5": (chair, dict(knn=5)),
        "chair_radius": (chair, dict(knn=None, radius=0.05)),
        "chair_hybrid": (chair, dict(knn=30, radius=0.05)),
        "chair_hybrid_small": (chair, dict(knn=10, radius=0.01)),          # many points with < 3 neighbours
        "frag_knn30": (frag, dict(knn=30)),
        "frag_hybrid_keep_sign": (frag, dict(knn=30, radius=0.1, normals=frag_n)),
        "frag_knn_keep_sign": (frag, dict(knn=20, normals=-frag_n)),
        "dup_knn10": (dup, dict(knn=10)),
        "line_knn8": (line, dict(knn=8)),
        "line_knn8_keep": (line, dict(knn=8, normals=np.tile([0.0, 1.0, 0.0], (40, 1)))),
        "chair_knn2": (chair[:50], dict(knn=2)),                           # fewer than 3 neighbours everywhere
    }
    for name, (pts, kw) in cases.items():
        out[name] = ref.estimate_normals(pts, **kw)
        print("normals %-24s n=%d  (0,0,1): %d" % (name, len(pts), int((out[name] == [0, 0, 1]).all(1).sum())))
    np.savez_compressed(os.path.join(HERE, "normals.npz"), **out)


def main():
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    ref = Ref()
    if len(sys.argv) > 1 and sys.argv[1] == "voxel":
        return gen_voxel(ref)
    if len(sys.argv) > 1 and sys.argv[1] == "mesh":
        return gen_mesh(ref)
    if len(sys.argv) > 1 and sys.argv[1] == "io":
        return gen_io(ref)
    if len(sys.argv) > 1 and sys.argv[1] == "pcd":
        return gen_pcd(ref)
    if len(sys.argv) > 1 and sys.argv[1] == "normals":
        return gen_normals(ref)
    V, F = load_obj(REF + "/misc/hermanmiller_aeron.obj")

    # ---- C1/C2: chair CAD 5k samples -> 20k partial noisy scan ------------
    src = f32(sample_mesh(V, F, 5000, 11)).astype(np.float64)
    dense = sample_mesh(V, F, 60000, 12)
    view = np.array([math.cos(0.6), 0.25, math.sin(0.6)])
    keep = np.argsort(-(dense @ view))[:20000]          # the half facing the "camera"
    keep.sort()
    rng = np.random.Generator(np.random.Philox(13))
    scan = dense[keep] + rng.standard_normal((20000, 3)) * 0.002
    T_true = make_T(rot_y(math.radians(10.0)), [0.03, 0.0, -0.02])
    tgt = f32(scan @ T_true[:3, :3].T + T_true[:3, 3]).astype(np.float64)
    tr, last = trace(ref, src, tgt, 0.075, np.eye(4), 20)
    np.savez_compressed(os.path.join(HERE, "chair_5k_20k.npz"), src=f32(src), tgt=f32(tgt),
                        radius=0.075, init=np.eye(4), trace=tr, final_idx=last.idx,
                        T_true=T_true)
    print("chair_5k_20k: K=%d fitness=%.4f rmse=%.5f" % (last.k, last.fitness, last.rmse))

    # same scene 3 m away from the origin (fp32 cancellation stress)
    off = np.array([3.0, -1.0, 2.5])
    src_o = f32(src + off).astype(np.float64)
    tgt_o = f32(tgt + off).astype(np.float64)
    init_o = make_T(rot_y(math.radians(2.0)), [0.005, 0.0, 0.004])
    tr_o, last_o = trace(ref, src_o, tgt_o, 0.075, init_o, 20)
    np.savez_compressed(os.path.join(HERE, "chair_offset3m.npz"), src=f32(src_o), tgt=f32(tgt_o),
                        radius=0.075, init=init_o, trace=tr_o[[0, 1, 5, 10, 20]],
                        trace_iters=np.array([0, 1, 5, 10, 20]), final_idx=last_o.idx)

    # default criteria (1e-6, 1e-6, 30): termination semantics
    term = ref.registration_icp(src, tgt, 0.075, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6)
    # with scaling (7-DoF Umeyama)
    src_s = f32(src * 1.03).astype(np.float64)
    sc = ref.registration_icp(src_s, tgt, 0.075, max_iter=15, rel_fitness=0, rel_rmse=0,
                              with_scaling=True)

    # ---- orientation-constrained sweep: 24 yaw inits (src/annotation.cpp:29-64)
    model = f32(sample_mesh(V, F, 4000, 21)).astype(np.float64)
    scene_dense = sample_mesh(V, F, 30000, 22)
    keep = np.argsort(-(scene_dense @ view))[:8000]
    keep.sort()
    rng = np.random.Generator(np.random.Philox(23))
    T_scene = make_T(rot_y(math.radians(97.0)), [0.01, 0.0, -0.015])
    scene = scene_dense[keep] + rng.standard_normal((8000, 3)) * 0.002
    scene = f32(scene @ T_scene[:3, :3].T + T_scene[:3, 3]).astype(np.float64)
    sweep_T, sweep_k, sweep_fit, sweep_rmse = [], [], [], []
    for i in range(24):
        a = 2 * math.pi / 24 * i
        r = ref.registration_icp(model, scene, 0.02, init=make_T(rot_y(a), [0, 0, 0]), max_iter=30,
                                 rel_fitness=1e-6, rel_rmse=1e-6)
        sweep_T.append(r.T); sweep_k.append(r.k); sweep_fit.append(r.fitness); sweep_rmse.append(r.rmse)
    sweep_k = np.array(sweep_k)
    best = int(np.argmax(sweep_k))   # np.argmax = first maximum = "strictly greater" rule
    np.savez_compressed(os.path.join(HERE, "yaw_sweep.npz"), model=f32(model), scene=f32(scene),
                        radius=0.02, level=24, T=np.array(sweep_T), k=sweep_k,
                        fitness=np.array(sweep_fit), rmse=np.array(sweep_rmse), best=best,
                        T_scene=T_scene)
    print("yaw sweep: best level %d K=%d (K range %d..%d)" % (best, sweep_k[best], sweep_k.min(), sweep_k.max()))

    # ---- real scan fragments with normals (Open3D TestData/Feature) --------
    a_xyz, a_n = load_pcd_xyz_normals(O3D_DATA + "/Feature/cloud_bin_0.pcd")
    b_xyz, b_n = load_pcd_xyz_normals(O3D_DATA + "/Feature/cloud_bin_1.pcd")
    a_xyz, a_n, b_xyz, b_n = [f32(x).astype(np.float64) for x in (a_xyz, a_n, b_xyz, b_n)]
    init_f = make_T(rot_y(0.02), [0.01, -0.005, 0.0])
    p2p, _ = trace(ref, a_xyz, b_xyz, 0.25, init_f, 10)
    p2l_rows = []
    for m in range(11):
        r = ref.registration_icp(a_xyz, b_xyz, 0.25, init=init_f, max_iter=m, rel_fitness=0,
                                 rel_rmse=0, estimator=EST_POINT_TO_PLANE, src_normals=a_n,
                                 tgt_normals=b_n)
        p2l_rows.append(np.concatenate([r.T.ravel(), [r.fitness, r.rmse, r.k]]))
    np.savez_compressed(os.path.join(HERE, "fragments.npz"), src=f32(a_xyz), src_normals=f32(a_n),
                        tgt=f32(b_xyz), tgt_normals=f32(b_n), radius=0.25, init=init_f,
                        trace_p2p=p2p, trace_p2plane=np.array(p2l_rows))
    print("fragments: p2p K=%d  p2plane K=%d" % (p2p[-1][18], p2l_rows[-1][18]))

    # ---- estimator-level vectors ------------------------------------------
    rng = np.random.default_rng(31)
    K = 700
    corr = np.stack([rng.integers(0, len(a_xyz), K), rng.integers(0, len(b_xyz), K)], 1).astype(np.int32)
    est = dict(corr=corr)
    est["rmse_p2p"] = ref.compute_rmse(a_xyz, b_xyz, corr)
    est["T_p2p"] = ref.compute_transformation(a_xyz, b_xyz, corr)
    est["T_p2p_scaled"] = ref.compute_transformation(a_xyz, b_xyz, corr, with_scaling=True)
    est["T_p2plane"] = ref.compute_transformation(a_xyz, b_xyz, corr, estimator=EST_POINT_TO_PLANE,
                                                  tgt_normals=b_n)
    est["T_empty"] = ref.compute_transformation(a_xyz, b_xyz, corr[:0])
    est["rmse_empty"] = ref.compute_rmse(a_xyz, b_xyz, corr[:0])
    # near-aligned correspondences (small residuals) for the 6x6 path
    idx = rng.integers(0, len(b_xyz), K)
    Tn = make_T(rot_y(0.01), [0.002, -0.001, 0.003])
    near = b_xyz[idx] @ np.linalg.inv(Tn)[:3, :3].T + np.linalg.inv(Tn)[:3, 3]
    near = f32(near + rng.standard_normal(near.shape) * 1e-3).astype(np.float64)
    corr_n = np.stack([np.arange(K), idx], 1).astype(np.int32)
    est["near_src"] = f32(near)
    est["near_corr"] = corr_n
    est["near_T_p2p"] = ref.compute_transformation(near, b_xyz, corr_n)
    est["near_T_p2plane"] = ref.compute_transformation(near, b_xyz, corr_n,
                                                       estimator=EST_POINT_TO_PLANE, tgt_normals=b_n)
    # 6x6 solves: a regular system, and a singular one that must be rejected
    A = rng.standard_normal((6, 6)); A = A @ A.T + 6 * np.eye(6); b6 = rng.standard_normal(6) * 0.01
    ok, X = ref.solve_jacobian_system(A, b6)
    As = A.copy(); As[5] = As[4]; As[:, 5] = As[:, 4]
    ok_s, X_s = ref.solve_jacobian_system(As, b6)
    est.update(jtj=A, jtr=b6, solve_ok=ok, solve_T=X, jtj_singular=As, solve_singular_ok=ok_s,
               solve_singular_T=X_s)
    xs = rng.standard_normal((8, 6)) * np.array([1.5, 0.7, 2.0, 1, 1, 1])
    est["euler_x"] = xs
    est["euler_T"] = np.array([ref.vector6d_to_matrix4d(x) for x in xs])
    est["termination_T"] = term.T
    est["termination"] = np.array([term.fitness, term.rmse, term.k])
    est["scaled_src"] = f32(src_s)
    est["scaled_T"] = sc.T
    est["scaled"] = np.array([sc.fitness, sc.rmse, sc.k])
    np.savez_compressed(os.path.join(HERE, "estimators.npz"), **est)

    # ---- edge cases ---------------------------------------------------------
    e_src = f32(sample_mesh(V, F, 300, 41)).astype(np.float64)
    e_tgt = f32(sample_mesh(V, F, 500, 42)).astype(np.float64)
    dup = np.concatenate([e_tgt, e_tgt[:50]], 0)        # exact duplicate targets
    edge = dict(src=f32(e_src), tgt=f32(e_tgt), tgt_dup=f32(dup))
    cases = {
        "none": (e_src + 50.0, e_tgt, 0.01, 5),          # no correspondences at all
        "tiny_radius": (e_src, e_tgt, 1e-4, 5),
        "dup": (e_src, dup, 0.05, 8),
        "one_src": (e_src[:1], e_tgt, 0.5, 4),
        "one_tgt": (e_src, e_tgt[:1], 2.0, 4),
        "huge_radius": (e_src, e_tgt, 1e3, 6),
        "zero_iter": (e_src, e_tgt, 0.05, 0),
    }
    for name, (s, t, r, m) in cases.items():
        res = ref.registration_icp(s, t, r, max_iter=m, rel_fitness=0, rel_rmse=0)
        edge[name + "_T"] = res.T
        edge[name + "_frk"] = np.array([res.fitness, res.rmse, res.k])
        edge[name + "_args"] = np.array([r, m])
    bad = ref.registration_icp(e_src, e_tgt, 0.0, init=make_T(rot_y(0.3), [1, 2, 3]), max_iter=5)
    edge["bad_radius_T"] = bad.T
    edge["bad_radius_frk"] = np.array([bad.fitness, bad.rmse, bad.k])
    noN = ref.registration_icp(e_src, e_tgt, 0.05, init=make_T(rot_y(0.3), [1, 2, 3]), max_iter=5,
                               estimator=EST_POINT_TO_PLANE)
    edge["plane_without_normals_T"] = noN.T
    ev = ref.evaluate_registration(e_src, e_tgt, 0.05, make_T(rot_y(0.05), [0.01, 0, 0]))
    edge["evaluate_frk"] = np.array([ev.fitness, ev.rmse, ev.k])
    edge["evaluate_idx"] = ev.idx
    edge["evaluate_T"] = make_T(rot_y(0.05), [0.01, 0, 0])
    np.savez_compressed(os.path.join(HERE, "edge_cases.npz"), **edge)

    # ---- Open3D known-answer unit tests (literals held by the reference tests) --
    pts = glibc_rand_points(100)
    ka = dict(rand_points=pts)
    # O3D/UnitTest/Core/Geometry/PointCloud.cpp:172-232 (PointCloud.Transform)
    ka["transform_T"] = np.array([[0.10, 0.20, 0.30, 0.40], [0.50, 0.60, 0.70, 0.80],
                                  [0.90, 0.10, 0.11, 0.12], [0.13, 0.14, 0.15, 0.16]])
    ka["transform_ref_points"] = np.array([
        [398.225124, 1205.693071, 881.868153], [321.838886, 1085.294390, 831.611417],
        [270.900608, 823.791432, 409.198658], [339.937683, 1004.432856, 615.608467],
        [425.227547, 1157.793590, 484.511386], [434.350931, 1342.432421, 967.169396],
        [140.844202, 447.193004, 190.052250], [293.388019, 767.506059, 320.900694],
        [135.193922, 410.559494, 195.502569], [276.542855, 807.338946, 221.948633]])
    ka["transform_ref_normals"] = np.array([
        [397.825124, 1204.893071, 881.748153], [321.438886, 1084.494390, 831.491417],
        [270.500608, 822.991432, 409.078658], [339.537683, 1003.632856, 615.488467],
        [424.827547, 1156.993590, 484.391386], [433.950931, 1341.632421, 967.049396],
        [140.444202, 446.393004, 189.932250], [292.988019, 766.706059, 320.780694],
        [134.793922, 409.759494, 195.382569], [276.142855, 806.538946, 221.828633]])
    # O3D/UnitTest/Core/Geometry/PointCloud.cpp:1074-1111
    ka["nn_distance_ref"] = np.array([
        155.013456, 126.672493, 114.606722, 190.747153, 133.079840, 121.137276, 106.805907,
        226.190750, 131.745147, 172.069584, 247.822223, 119.390962, 21.209580, 68.624498,
        136.386737, 149.981320, 206.445708, 191.876431, 140.127314, 131.657386, 183.471289,
        221.094822, 178.447628, 126.081556, 29.338770, 111.453558, 102.236849, 304.969947,
        40.823263, 227.787078, 169.129676, 197.146871, 167.494524, 174.795150, 142.910946,
        263.053174, 122.803815, 238.740548, 116.243401, 180.230879, 91.863637, 96.241462,
        24.547707, 174.705689, 65.612463, 148.994593, 158.758879, 345.655903, 251.182091,
        182.235820])
    # sanity: the compiled reference reproduces its own literals from these inputs
    p10, n10 = ref.transform_points(pts[:10], ka["transform_T"], normals=pts[:10])
    assert np.abs(p10 - ka["transform_ref_points"]).max() < 1e-6
    assert np.abs(n10 - ka["transform_ref_normals"]).max() < 1e-6
    d = ref.nn_distance(pts[:50], pts[50:100])
    assert np.abs(d - ka["nn_distance_ref"]).max() < 1e-6
    np.savez_compressed(os.path.join(HERE, "open3d_known_answers.npz"), **ka)

    # ---- SO(3): core/rodrigues.h outputs -----------------------------------
    rng = np.random.default_rng(51)
    ws = np.concatenate([rng.standard_normal((12, 3)), rng.standard_normal((3, 3)) * 1e-10,
                         np.zeros((1, 3)), rng.standard_normal((2, 3)) * 1e-4])
    Rs, dRs, w2, dws, hats = [], [], [], [], []
    for w in ws:
        R, dR = ref.rodrigues(w)
        wb, dw = ref.invrodrigues(R)
        Rs.append(R); dRs.append(dR); w2.append(wb); dws.append(dw); hats.append(ref.hat(w))
    np.savez_compressed(os.path.join(HERE, "rodrigues.npz"), w=ws, R=np.array(Rs),
                        dR_dw=np.array(dRs), w_back=np.array(w2), dw_dR=np.array(dws),
                        hat=np.array(hats))
    gen_voxel(ref)
    gen_mesh(ref)
    gen_io(ref)
    gen_pcd(ref)
    gen_normals(ref)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
