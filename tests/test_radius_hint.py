"""visma_icp_set_radius_hint: told the radius of the coming registration, the upload builds the search structure on the
stream while the host stages the source.  Results never depend on the hint (right, wrong, absent, or the previous
registration's radius remembered by the context)."""
import numpy as np
import pytest


def test_abi_exports_the_hint(lib):
    assert hasattr(lib.load(), "visma_icp_set_radius_hint")


@pytest.mark.gpu
@pytest.mark.parametrize("ns,nt", [(5000, 20000), (60000, 400000)])
def test_results_do_not_depend_on_the_hint(lib, ns, nt):
    from visma_amd import _lib, synth
    src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
    src2, tgt2, _, _ = synth.make_pair(ns // 2 + 7, nt // 2 + 3, seed_t=5, seed_s=6, motion="radius")

    def register(c, s, t, radius):
        c.set_clouds_f64(s, t)
        res = c.run(None, radius, 12, 0.0, 0.0)
        return res.transformation_.copy(), res.num_correspondences, c.correspondence_index().copy()

    plain = _lib.Context(0)
    want = register(plain, src, tgt, r)
    want2 = register(_lib.Context(0), src2, tgt2, 1.5 * r)
    for hint in (r, 3.0 * r, 1e-9, 0.0):
        c = _lib.Context(0)
        c.set_radius_hint(hint)
        got = register(c, src, tgt, r)
        assert got[1] == want[1] and np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]), hint
        # the context now remembers r (with a hint still set, the hint wins): other clouds, another radius
        got2 = register(c, src2, tgt2, 1.5 * r)
        assert got2[1] == want2[1] and np.array_equal(got2[0], want2[0]) and np.array_equal(got2[2], want2[2]), hint
        # brute force asked for after a grid was prepared: the prepared grid is simply not used
        c.set_nn_mode(_lib.NN_BRUTE)
        got3 = register(c, src[:3000], tgt[:20000], r)
        c.set_nn_mode(_lib.NN_GRID)
        c.set_radius_hint(0.0)
        got4 = register(c, src[:3000], tgt[:20000], r)
        assert got3[1] == got4[1] and np.array_equal(got3[2], got4[2])
        c.close()
    # point-to-plane after a hinted upload (normals arrive after the grid exists)
    c = _lib.Context(0)
    nrm = c.estimate_normals(tgt, knn=12, radius=4 * r)
    outs = []
    for hint in (0.0, r):
        c.set_radius_hint(hint)
        c.set_clouds_f64(src, tgt)
        c.set_target_normals_f64(nrm)
        res = c.run_point_to_plane(None, r, 6, 0.0, 0.0)
        outs.append(res.transformation_.copy())
    assert np.array_equal(outs[0], outs[1])
    c.close()
    with pytest.raises(_lib.IcpError):
        plain.set_radius_hint(float("nan"))
    plain.close()
