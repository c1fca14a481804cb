#!/usr/bin/env python3
"""Generate tests/golden/c4_ref.npz: BASELINE config 4 at FULL size (262,144 -> 4,194,304 points)
run through the COMPILED REFERENCE (oracle/_ref: Open3D's own RegistrationICP, Registration.cpp:141-186,
with its KDTreeFlann), so that `pytest -m gpu` compares the default exact kernel at the size bench.py
times with the reference's own output instead of a self-report.

Stored: the recipe (the clouds come back from visma_amd.synth, a counter-based Philox stream that is
identical on every machine; a checksum of the generated inputs guards that claim), and what the
reference returned for
  * one evaluation pass at a non-trivial pose (`EvaluateRegistration`, Registration.cpp:98-116):
    K, fitness, rmse, checksums of `correspondence_set_`;
  * `RegistrationICP` from identity, 10 iterations, criteria (0, 0, 10): final transformation, K, fitness,
    rmse, checksums of the final `correspondence_set_`.
A correspondence checksum is (sum of target indices, sum of (i + 1) * index mod 2^61 - 1): a swap of two
partners or a single changed index moves the second one.

    python tests/golden/gen_c4.py [--partial | --literal]
                                                   (--partial: tests/golden/c4_partial_ref.npz, the whole model against
                                                    a scan of half of its surface: fitness ~ 0.5;
                                                    --literal: tests/golden/c4_literal_ref.npz, SURVEY 8d's literal
                                                    ground truth with r = 0.15 m: the large-radius regime)
Runs in THIS container only (needs oracle/_ref; ~1 minute on 8 cores); the .npz travels."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from visma_amd import synth  # noqa: E402

NS, NT = 262144, 4194304
ITERS = 10
M61 = (1 << 61) - 1


def clouds():
    """The C4 clouds of bench.py / tests/test_gpu_fullsize.py."""
    return synth.make_pair(NS, NT, motion="radius")


def partial_clouds():
    """The partial-overlap variant of C4 (bench.py: `partial_overlap`): the whole model against half of its surface."""
    return synth.make_partial_pair(NS, NT, overlap=0.5)


def literal_clouds():
    """C4's sizes under SURVEY 8d's LITERAL ground truth (T_gt = R_y(5 deg) R_x(1 deg), t = (0.02, -0.01, 0.015)) with the
    radius that motion needs (bench.py: `literal_T_gt`; the regime of visma_amd/csrc/grid_ring.hip: 3,200 points per
    radius-sized cell)."""
    src, tgt, T_gt, _ = synth.make_pair(NS, NT, motion="fixed")
    return src, tgt, T_gt, 0.15


def eval_pose(r):
    """The pose of the single evaluation pass (a fraction of the radius away from identity)."""
    return synth.make_T(synth.rot_y(0.3 * r), [0.2 * r, 0, -0.1 * r])


def checksum(idx):
    """(sum of matched target indices, position-weighted sum mod 2^61-1, K) of a per-source index array (-1 = none)."""
    idx = np.asarray(idx, dtype=np.int64)
    m = idx >= 0
    return int(idx[m].sum()), _weighted(idx, m), int(m.sum())


def _weighted(idx, m):
    # exact modular arithmetic in numpy: split the weights so that products stay below 2^63
    i = np.nonzero(m)[0].astype(np.uint64) + np.uint64(1)
    v = idx[m].astype(np.uint64)
    lo = (i & np.uint64(0xFFFF)) * v                 # < 2^16 * 2^23
    hi = (i >> np.uint64(16)) * v                    # < 2^7 * 2^23 at these sizes
    total = (int(lo.sum(dtype=np.uint64)) + (int(hi.sum(dtype=np.uint64)) << 16)) % M61
    return total


def input_checksum(a):
    return int(np.ascontiguousarray(a).view(np.uint64).sum(dtype=np.uint64))


def main():
    from oracle.oracle import Ref
    ref = Ref()
    if "--partial" in sys.argv:
        src, tgt, T_gt, r = partial_clouds()
        name = "c4_partial_ref.npz"
    elif "--literal" in sys.argv:
        src, tgt, T_gt, r = literal_clouds()
        name = "c4_literal_ref.npz"
    else:
        src, tgt, T_gt, r = clouds()
        name = "c4_ref.npz"
    out = {"ns": NS, "nt": NT, "radius": r, "iters": ITERS,
           "src_checksum": np.uint64(input_checksum(src)), "tgt_checksum": np.uint64(input_checksum(tgt))}
    t0 = time.time()
    # (--literal: a pose a tenth of the way to the ground truth -- most nearest neighbours centimetres away)
    T0 = eval_pose(r) if "--literal" not in sys.argv else synth.make_T(synth.rot_y(np.deg2rad(0.5)), [0.002, -0.001, 0.0015])
    e = ref.evaluate_registration(src, tgt, r, T0)
    s1, s2, k = checksum(e.idx)
    assert k == e.k
    out.update(eval_T=T0, eval_k=e.k, eval_fitness=e.fitness, eval_rmse=e.rmse, eval_sum=np.int64(s1), eval_wsum=np.int64(s2))
    print("evaluate: K=%d fitness=%.6f rmse=%.9f  (%.1f s)" % (e.k, e.fitness, e.rmse, time.time() - t0))
    t0 = time.time()
    w = ref.registration_icp(src, tgt, r, init=np.eye(4), max_iter=ITERS, rel_fitness=0.0, rel_rmse=0.0)
    s1, s2, k = checksum(w.idx)
    assert k == w.k
    out.update(ref_T=np.asarray(w.T).reshape(4, 4), ref_k=w.k, ref_fitness=w.fitness, ref_rmse=w.rmse,
               ref_sum=np.int64(s1), ref_wsum=np.int64(s2))
    print("icp x%d: K=%d fitness=%.6f rmse=%.9f  (%.1f s)" % (ITERS, w.k, w.fitness, w.rmse, time.time() - t0))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name)
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
