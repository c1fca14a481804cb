"""tools/edge_probe.py inside the suite: every entry point of the wrapper with empty, one-point and degenerate inputs (empty
source / target / both under every search mode through the per-pass API, host loop, device loop, yaw sweep and fixed
iterations; voxel grid, normals, mesh sampling and distances on nothing; batches with empty problems; radii 0, nan, inf, 1e30,
1e-9; non-finite points; zero iterations; the ring search asked for on all of it) -- a value or a clean error from each,
never a crash, a hang or a launch the device refuses."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_every_entry_point_survives_empty_and_degenerate_inputs(lib):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "edge_probe.py")], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    out = p.stdout
    assert p.returncode == 0 and out.rstrip().endswith("done"), out[-2000:] + p.stderr[-2000:]
    bad = [ln for ln in out.splitlines() if ln.startswith(("error", "PYERR"))]
    assert not bad, "\n".join(bad)
    assert sum(ln.startswith("ok") for ln in out.splitlines()) >= 40
    # what the empty cases must say
    for ln in out.splitlines():
        if "source empty" in ln or "target empty" in ln:
            assert "('K', np.float64(0.0))" in ln and "('run', 0)" in ln and "('sweep', 0)" in ln, ln
