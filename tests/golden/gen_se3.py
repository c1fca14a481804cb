#!/usr/bin/env python3
"""Golden vectors of the reference's SE3Type / SO3Type (core/se3.h:10-169), from the header itself compiled into
oracle/_ref (oracle/ref_se3.cpp: the image's clang, -fdelayed-template-parsing -- the header does not build with g++).
Run where /root/reference exists:   python tests/golden/gen_se3.py   -> tests/golden/se3.npz (inputs and outputs only)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import Ref  # noqa: E402


def rot(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    return q * np.sign(np.linalg.det(q))


def main():
    ref = Ref()
    rng = np.random.default_rng(2026)
    n = 40
    g = np.zeros((n, 3, 4)); h = np.zeros((n, 3, 4))
    for k in range(n):
        g[k, :, :3] = rot(rng); h[k, :, :3] = rot(rng)
        g[k, :, 3] = rng.standard_normal(3) * 10.0 ** rng.uniform(-3, 2)
        h[k, :, 3] = rng.standard_normal(3) * 10.0 ** rng.uniform(-3, 2)
    g[0] = np.hstack([np.eye(3), np.zeros((3, 1))])                       # the identity
    h[1, :, :3] = g[1, :, :3].T; h[1, :, 3] = -g[1, :, :3].T @ g[1, :, 3]  # an inverse pair
    v = rng.standard_normal((n, 3)) * 10.0 ** rng.uniform(-2, 2, (n, 1))
    w = np.concatenate([rng.standard_normal((n - 6, 3)), rng.standard_normal((3, 3)) * 1e-10, np.zeros((1, 3)),
                        rng.standard_normal((2, 3)) * 1e-4])
    axis = rng.standard_normal((n, 3)) * 10.0 ** rng.uniform(-2, 2, (n, 1))
    angle = rng.uniform(-3.0, 3.0, n)
    out = dict(g=g, h=h, v=v, w=w, axis=axis, angle=angle,
               gh=np.array([ref.se3_compose(a, b) for a, b in zip(g, h)]),
               gv=np.array([ref.se3_act(a, x) for a, x in zip(g, v)]),
               ginv=np.array([ref.se3_inv(a) for a in g]),
               g44=np.array([ref.se3_matrix(a) for a in g]),
               expw=np.array([ref.so3_exp(x) for x in w]))
    out["logR"] = np.array([ref.so3_log(R) for R in out["expw"]])
    out["axis_angle"] = np.array([ref.so3_axis_angle(a, t) for a, t in zip(axis, angle)])
    # sanity of the harness itself: the reference's own classes satisfy their axioms on these inputs
    for k in range(n):
        e = ref.se3_compose(g[k], out["ginv"][k])
        assert np.allclose(e, np.hstack([np.eye(3), np.zeros((3, 1))]), atol=1e-12 * (1 + np.abs(g[k]).max()))
        assert np.allclose(ref.se3_act(out["gh"][k], v[k]), ref.se3_act(g[k], ref.se3_act(h[k], v[k])), rtol=1e-12, atol=1e-10)
    np.savez_compressed(os.path.join(HERE, "se3.npz"), **out)
    print("wrote se3.npz:", {k: a.shape for k, a in out.items()})


if __name__ == "__main__":
    main()
