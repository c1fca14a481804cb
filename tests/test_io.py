"""On-disk formats either side of the ICP path (include/visma_io.h).

Host code: runs without a GPU.  Every expected array in golden/io.npz is an OUTPUT of the
reference's own readers (open3d::ReadPointCloudFromPLY / ReadTriangleMeshFromPLY on rply,
igl::readOBJ) for the files under golden/io/ -- written by tests/golden/gen_golden.py, except
cube.ply, a data file of the reference.  The bar is bit-exact: reading is integer / byte work
plus exact widening to double.
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
D = os.path.join(HERE, "golden", "io")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(HERE, "golden", "io.npz")))


@pytest.fixture(scope="module")
def io(lib):
    from visma_amd import _lib
    return _lib


PLY_OK = ["cube.ply", "gen_le.ply", "gen_be.ply", "gen_ascii.ply", "gen_points.ply"]


@pytest.mark.parametrize("name", PLY_OK)
def test_ply_matches_the_reference_readers(io, gold, name):
    key = name.replace(".", "_")
    assert gold[key + "_ok"].all()
    got = io.read_ply(os.path.join(D, name))
    for k in ("xyz", "normals", "colors"):
        want = gold[key + "_" + k]
        assert got[k].shape == want.shape, k
        assert np.array_equal(got[k], want), k           # bit-exact (NaN-free by construction)
    assert np.array_equal(got["faces"], gold[key + "_faces"])


def test_ply_semantics_spelled_out(io):
    c = io.read_ply(os.path.join(D, "gen_le.ply"))
    assert c["xyz"].shape == (300, 3) and c["normals"].shape == (300, 3) and c["colors"].shape == (300, 3)
    assert c["colors"].min() >= 0 and c["colors"].max() <= 1.0                 # value / 255.0
    assert np.array_equal(c["xyz"], c["xyz"].astype(np.float32).astype(np.float64))   # float widened exactly
    assert c["faces"].shape == (126, 3)                  # quads give their first 3 corners; the 2-gon keeps a 0
    a = io.read_ply(os.path.join(D, "gen_ascii.ply"))
    b = io.read_ply(os.path.join(D, "gen_be.ply"))
    assert np.array_equal(a["xyz"][:, 0], b["xyz"][:, 0])    # %.17g text of a "float" property keeps the double
    assert b["colors"].max() > 1.0                         # ushort colours: still value / 255.0, as the reference
    p = io.read_ply(os.path.join(D, "gen_points.ply"))
    assert p["normals"].shape == (0, 3) and p["colors"].shape == (0, 3) and p["faces"].shape == (0, 3)


@pytest.mark.parametrize("name", ["bad_truncated.ply", "bad_novertex.ply", "does_not_exist.ply", "tri.obj",
                                  "bad_hugecount.ply", "bad_hugecount_ascii.ply", "bad_ny_only.ply", "bad_green_only.ply"])
def test_ply_failures_are_reported(io, gold, name):
    key = name.replace(".", "_") + "_ok"
    if key in gold and name.endswith(".ply"):
        assert not gold[key].any()                       # the reference fails on it too
    with pytest.raises(io.IoError):
        io.read_ply(os.path.join(D, name))               # (an OBJ file is "not a PLY file")


def test_repeated_face_element_keeps_the_first(io):
    c = io.read_ply(os.path.join(D, "two_faces.ply"))
    assert c["faces"].shape == (1, 3) and c["xyz"].shape == (3, 3)


OBJ_OK = ["tri.obj", "mixed_syntax.obj", "negative.obj", "quads.obj", "colors.obj"]


@pytest.mark.parametrize("name", OBJ_OK)
def test_obj_matches_igl(io, gold, name):
    key = name.replace(".", "_")
    V, F = io.read_obj(os.path.join(D, name))
    assert np.array_equal(V, gold[key + "_V"][:, :3])    # LoadMesh keeps leftCols(3) (core/utils.cpp:132)
    assert np.array_equal(F, gold[key + "_F"])
    if name == "quads.obj":
        assert F.shape[1] == 4 and gold[key + "_V"].shape[1] == 4
    if name == "colors.obj":
        assert gold[key + "_V"].shape[1] == 6


@pytest.mark.parametrize("name", ["bad_mixed_faces.obj", "bad_short_vertex.obj", "bad_face_token.obj", "nope.obj"])
def test_obj_failures_are_reported(io, gold, name):
    key = name.replace(".", "_") + "_ok"
    if key in gold:
        assert not gold[key].any()
    with pytest.raises(io.IoError):
        io.read_obj(os.path.join(D, name))


def test_obj_feeds_the_mesh_fixture(io):
    """The chair of golden/mesh.npz written as OBJ and read back: same arrays."""
    m = np.load(os.path.join(HERE, "golden", "mesh.npz"))
    import tempfile
    with tempfile.TemporaryDirectory() as t:
        path = os.path.join(t, "chair.obj")
        with open(path, "w") as f:
            for v in m["V"]:
                f.write("v %r %r %r\n" % tuple(float(x) for x in v))
            for tri in m["F"]:
                f.write("f %d %d %d\n" % tuple(int(i) + 1 for i in tri))
        V, F = io.read_obj(path)
    assert np.array_equal(V, m["V"]) and np.array_equal(F, m["F"])


# ---------------------------------------------------------------- PCD
PCD_GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pcd.npz"))
PCD_OK = ["gen_ascii.pcd", "gen_binary.pcd", "gen_int.pcd", "gen_wide.pcd", "ref_compressed.pcd",
          "ref_compressed_xyz.pcd", "ref_ascii.pcd"]


@pytest.mark.parametrize("name", PCD_OK)
def test_pcd_matches_the_compiled_reference_bit_for_bit(io, name):
    """tests/golden/pcd.npz = outputs of open3d::ReadPointCloudFromPCD (FilePCD.cpp:727-760) built into oracle/_ref."""
    key = name.replace(".", "_")
    assert PCD_GOLD[key + "_ok"][0]
    c = io.read_pcd(os.path.join(D, name))
    for k in ("xyz", "normals", "colors"):
        want = PCD_GOLD[key + "_" + k]
        assert c[k].shape == want.shape, (name, k, c[k].shape, want.shape)
        assert np.array_equal(c[k], want, equal_nan=True), (name, k)
    assert c["faces"].shape == (0, 3)


def test_pcd_nan_rule_is_the_references(io):
    """Rows with a NaN x or y go; a NaN z stays (the reference tests x twice, FilePCD.cpp:521-523)."""
    c = io.read_pcd(os.path.join(D, "gen_ascii.pcd"))
    assert len(c["xyz"]) == 257 - 2 and np.isnan(c["xyz"][:, 2]).sum() == 1


@pytest.mark.parametrize("name", ["bad_truncated.pcd", "bad_nofields.pcd", "bad_nopoints.pcd", "bad_sizecount.pcd",
                                  "bad_lzf.pcd", "does_not_exist.pcd", "tri.obj", "bad_hugepoints.pcd", "bad_count0.pcd"])
def test_pcd_failures_are_reported(io, name):
    key = name.replace(".", "_") + "_ok"
    if key in PCD_GOLD:
        assert not PCD_GOLD[key][0]                      # the reference fails on it too
    with pytest.raises(io.IoError):
        io.read_pcd(os.path.join(D, name))


REF_PCDS = ["/root/reference/thirdparty/Open3D/examples/TestData/fragment.pcd",
            "/root/reference/thirdparty/Open3D/examples/TestData/Feature/cloud_bin_0.pcd",
            "/root/reference/thirdparty/Open3D/examples/TestData/Feature/cloud_bin_1.pcd"]


@pytest.mark.parametrize("path", REF_PCDS)
def test_pcd_reference_data_files(io, path):
    """The reference's own real-scan fixtures, where the reference tree is present (this container)."""
    from oracle.oracle import Ref
    if not (os.path.exists(path) and Ref.available()):
        pytest.skip("reference tree / oracle/_ref not present")
    want = Ref().read_pcd_cloud(path)
    got = io.read_pcd(path)
    for k in ("xyz", "normals", "colors"):
        assert np.array_equal(got[k], want[k], equal_nan=True), k


# ---------------------------------------------------------------- pose files
def test_alignment_json_round_trip_and_key_order(io, tmp_path):
    rng = np.random.default_rng(4)
    poses = [("chair_%d" % k, rng.standard_normal((3, 4))) for k in (2, 0, 10, 1)]
    p = tmp_path / "alignment.json"
    io.write_alignment_json(p, poses)
    import json
    raw = json.load(open(p))                             # what any JSON reader sees: name -> 12 numbers, row by row
    assert np.array_equal(np.array(raw["chair_10"]).reshape(3, 4), poses[2][1])
    back = io.read_alignment_json(p)
    assert [b["name"] for b in back] == sorted(n for n, _ in poses)       # jsoncpp iterates keys in order
    for b in back:
        T = dict(poses)[b["name"]]
        assert np.array_equal(b["T"], T)                 # %.17g round-trips doubles exactly


def test_alignment_json_as_the_reference_tools_write_it(io, tmp_path):
    p = tmp_path / "a.json"
    p.write_text('{\n\t"swivel_chair_3" : \n\t[\n\t\t1.0, 0, 0, 0.5,\n\t\t0, 1e0, 0, -2,\n\t\t0, 0, 1, 3.25\n\t],\n'
                 '\t"a\\"b_0" : [0,1,2,3,4,5,6,7,8,9,10,11, 99]\n}\n')
    back = io.read_alignment_json(p)
    assert [b["name"] for b in back] == ['a"b_0', "swivel_chair_3"]
    assert np.array_equal(back[1]["T"], np.array([[1, 0, 0, 0.5], [0, 1, 0, -2], [0, 0, 1, 3.25]]))
    assert np.array_equal(back[0]["T"].reshape(12), np.arange(12))


def test_result_json_last_packet(io, tmp_path):
    import json
    pk = lambda s: [{"id": 7 + s, "status": 1, "model_name": "chair", "model_pose": list(range(s, s + 12))},
                    {"id": 9, "status": 0, "model_name": "table", "model_pose": [0.5] * 12}]
    p = tmp_path / "result.json"
    json.dump([pk(0), pk(100)], open(p, "w"))
    last = io.read_result_json(p)
    assert [o["id"] for o in last] == [107, 9] and last[0]["name"] == "chair" and last[1]["status"] == 0
    assert np.array_equal(last[0]["T"].reshape(12), np.arange(100, 112))
    first = io.read_result_json(p, 0)
    assert first[0]["id"] == 7


@pytest.mark.parametrize("body", ["", "[1, 2", '{"a": [1,2,3]}', '{"a": 5}', "nonsense"])
def test_bad_pose_files_fail(io, tmp_path, body):
    p = tmp_path / "bad.json"
    p.write_text(body)
    with pytest.raises(io.IoError):
        io.read_alignment_json(p)
