"""Seeded synthetic clouds for the ICP path.

S-surf(N, seed): N points, area-uniform on a closed analytic surface with
scan-like anisotropy -- the union of a torus (R=1, r=0.35, axis +Y) and three
axis-aligned boxes, all inside [-1.5, 1.5]^3 -- drawn with a counter-based
PRNG (numpy Philox), so the same (N, seed) gives the same cloud on every
machine.  The ICP source is an independent sample of the same surface,
pre-multiplied by T_gt^-1, so the registration has a known answer.
"""
import math

import numpy as np

TORUS_R, TORUS_r = 1.0, 0.35
# (centre, half-extent) of the three boxes
BOXES = (
    ((0.0, -0.55, 0.0), (0.45, 0.20, 0.45)),
    ((1.05, 0.45, -0.9), (0.30, 0.55, 0.25)),
    ((-0.95, 0.10, 1.0), (0.40, 0.30, 0.35)),
)


def _box_area(h):
    return 8.0 * (h[0] * h[1] + h[1] * h[2] + h[0] * h[2])


TORUS_AREA = 4.0 * math.pi ** 2 * TORUS_R * TORUS_r
SURFACE_AREA = TORUS_AREA + sum(_box_area(h) for _, h in BOXES)


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def make_T(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def T_gt():
    """Ground-truth motion source -> target used by every synthetic pair."""
    return make_T(rot_y(math.radians(5.0)) @ rot_x(math.radians(1.0)),
                  [0.02, -0.01, 0.015])


def _torus(rng, n):
    out = np.empty((n, 3))
    got = 0
    while got < n:
        m = int((n - got) * 1.6) + 16
        u = rng.random(m) * 2 * math.pi
        v = rng.random(m) * 2 * math.pi
        w = rng.random(m)
        keep = w * (TORUS_R + TORUS_r) <= (TORUS_R + TORUS_r * np.cos(v))
        u, v = u[keep], v[keep]
        k = min(len(u), n - got)
        rad = TORUS_R + TORUS_r * np.cos(v[:k])
        out[got:got + k, 0] = rad * np.cos(u[:k])
        out[got:got + k, 1] = TORUS_r * np.sin(v[:k])
        out[got:got + k, 2] = rad * np.sin(u[:k])
        got += k
    return out


def _box(rng, n, centre, h):
    h = np.asarray(h, dtype=np.float64)
    areas = np.array([h[1] * h[2], h[1] * h[2], h[0] * h[2], h[0] * h[2],
                      h[0] * h[1], h[0] * h[1]])
    face = rng.choice(6, size=n, p=areas / areas.sum())
    uv = rng.random((n, 2)) * 2.0 - 1.0
    out = np.empty((n, 3))
    axis = face // 2
    sign = np.where(face % 2 == 0, 1.0, -1.0)
    for a in range(3):
        b, c = (a + 1) % 3, (a + 2) % 3
        m = axis == a
        out[m, a] = sign[m] * h[a]
        out[m, b] = uv[m, 0] * h[b]
        out[m, c] = uv[m, 1] * h[c]
    return out + np.asarray(centre, dtype=np.float64)


def surface_points(n, seed):
    """n area-uniform points on the S-surf surface, float64 (n, 3)."""
    rng = np.random.Generator(np.random.Philox(seed))
    areas = np.array([TORUS_AREA] + [_box_area(h) for _, h in BOXES])
    part = rng.choice(len(areas), size=n, p=areas / areas.sum())
    out = np.empty((n, 3))
    m = part == 0
    out[m] = _torus(rng, int(m.sum()))
    for b, (c, h) in enumerate(BOXES):
        m = part == b + 1
        out[m] = _box(rng, int(m.sum()), c, h)
    return out


def default_radius(nt):
    """r = 3 * sqrt(area / NT), clamped to [0.005, 0.075]."""
    return float(min(0.075, max(0.005, 3.0 * math.sqrt(SURFACE_AREA / nt))))


def T_gt_scaled(radius):
    """A motion of about one search radius: the large-cloud workloads use it so
    that radius-limited ICP (r shrinks with the target density) still converges
    within the fixed iteration budget."""
    a = radius / 1.5
    return make_T(rot_y(a) @ rot_x(a / 5.0), np.array([0.6, -0.3, 0.45]) * radius)


def make_pair(ns, nt, seed_t=1234, seed_s=5678, noise=1e-3, offset=None, motion="fixed"):
    """Return (source, target, T_gt, radius); clouds are float32-rounded f64.

    target = S-surf(nt, seed_t) + N(0, noise); source = T_gt^-1 * S-surf(ns, seed_s).
    `offset` (3-vector) shifts the whole scene away from the origin.
    motion = "fixed": T_gt() (5 deg yaw, ~3 cm);  "radius": T_gt_scaled(radius).
    """
    tgt = surface_points(nt, seed_t)
    rng = np.random.Generator(np.random.Philox(seed_t + 1))
    tgt += rng.standard_normal(tgt.shape) * noise
    src = surface_points(ns, seed_s)
    T = T_gt() if motion == "fixed" else T_gt_scaled(default_radius(nt))
    if offset is not None:
        off = np.asarray(offset, dtype=np.float64)
        tgt += off
        src += off
        # keep T_gt the motion that maps the shifted source onto the shifted target
    Ti = np.linalg.inv(T)
    src = src @ Ti[:3, :3].T + Ti[:3, 3]
    src = src.astype(np.float32).astype(np.float64)
    tgt = tgt.astype(np.float32).astype(np.float64)
    return src, tgt, T, default_radius(nt)


def partial_surface_points(n, seed, keep=0.5):
    """n points of the part of the S-surf surface a scan from ONE side would see: the `keep` fraction of the surface
    (by area, approximately) on the near side of a slanted plane.  Drawn from the same stream as surface_points --
    the first n of an over-sampled cloud that lie on the kept side -- so (n, seed, keep) fixes the cloud."""
    normal = np.array([0.8, 0.1, 0.59])
    normal /= np.linalg.norm(normal)
    # the plane's offset for the requested share, from a fixed probe sample of the full surface
    probe = surface_points(200000, 99) @ normal
    cut = float(np.quantile(probe, keep))
    over = int(n / keep * 1.08) + 1024
    pts = surface_points(over, seed)
    kept = pts[(pts @ normal) <= cut]
    if len(kept) < n:
        raise RuntimeError("partial_surface_points: over-sampling too small")
    return kept[:n]


def make_partial_pair(ns, nt, overlap=0.5, seed_t=4321, seed_s=8765, noise=1e-3):
    """A CAD model against a PARTIAL scan of it (what the reference's callers register: fitness 0.37-0.62 in Open3D's
    own tutorial, docs/tutorial/Basic/icp_registration.rst:91,116): the source samples the whole surface, the target
    only the `overlap` share of it one side of a plane (nt points: denser, like a scan), 1 mm noise.  About
    (1 - overlap) of the source has no partner within the radius -- those queries list their cells against the radius
    every pass.  Same radius rule and ground-truth motion as make_pair(..., motion="radius").
    Returns (source, target, T_gt, radius); clouds are float32-rounded f64."""
    tgt = partial_surface_points(nt, seed_t, overlap)
    rng = np.random.Generator(np.random.Philox(seed_t + 1))
    tgt = tgt + rng.standard_normal(tgt.shape) * noise
    src = surface_points(ns, seed_s)
    T = T_gt_scaled(default_radius(nt))
    Ti = np.linalg.inv(T)
    src = src @ Ti[:3, :3].T + Ti[:3, 3]
    return (src.astype(np.float32).astype(np.float64), tgt.astype(np.float32).astype(np.float64), T, default_radius(nt))


def make_source(ns, nt, seed_s=5678, motion="radius"):
    """The source cloud make_pair(ns, nt, seed_s=seed_s, motion=motion) returns, without generating the target
    again (the ground-truth motion depends on nt through the radius only)."""
    src = surface_points(ns, seed_s)
    T = T_gt() if motion == "fixed" else T_gt_scaled(default_radius(nt))
    Ti = np.linalg.inv(T)
    src = src @ Ti[:3, :3].T + Ti[:3, 3]
    return src.astype(np.float32).astype(np.float64)


def rel_frobenius(A, B):
    """||A - B||_F / ||B||_F on 4x4 transforms (the parity metric)."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    return float(np.linalg.norm(A - B) / np.linalg.norm(B))
