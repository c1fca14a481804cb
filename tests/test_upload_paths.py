"""visma_icp_set_clouds_f64 has two upload paths on the HIP engine: clouds sent as the caller's f64 values and
expanded / Morton-ordered on the device (default), or packed and ordered on the host
(VISMA_ICP_RAW_UPLOAD_MIN above the cloud size).  Same correspondences, same transform to rounding (the two
Morton orders differ, so the f64 sums are taken in another order)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(ns, nt, stride, prec, raw, kind="f32exact"):
    env = dict(os.environ)
    env["VISMA_ICP_RAW_UPLOAD_MIN"] = "0" if raw else "2000000000"
    out = subprocess.run([sys.executable, os.path.join(HERE, "upload_worker.py"), str(ns), str(nt), str(stride), prec, kind],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("ns,nt,stride,prec,kind", [(5000, 20000, 3, "exact", "f32exact"), (30000, 150000, 5, "exact", "f32exact"),
                                                    (8000, 40000, 3, "f32", "f32exact"), (20000, 100000, 3, "exact", "f64"),
                                                    (60000, 2500000, 4, "exact", "mixed")])
def test_raw_and_host_packed_uploads_agree(lib, ns, nt, stride, prec, kind):
    """(the raw upload sends fp32-representable pieces as fp32 and sums the centroid while it stages them)"""
    a, b = _run(ns, nt, stride, prec, True, kind), _run(ns, nt, stride, prec, False, kind)
    assert a["mode"] == b["mode"] == prec
    assert a["k"] == b["k"] and a["idx_sum"] == b["idx_sum"] and a["idx_hash"] == b["idx_hash"]
    Ta, Tb = np.array(a["T"]), np.array(b["T"])
    assert np.linalg.norm(Ta - Tb) / np.linalg.norm(Tb) < (1e-12 if prec == "exact" else 1e-9)
    assert abs(a["rmse"] - b["rmse"]) < 1e-12
