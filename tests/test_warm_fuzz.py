"""tools/fuzz_warm_vs_serial.py inside the suite: hostile configurations (lattices on cell faces, duplicated points,
planes, elongated boxes, far offsets, extreme radii), thirteen passes each (incl. the decaying motions the certificate decides) -- the warm-started wave-cooperative search and
the lane-serial kernel agree bit for bit (indices, distances, the 38 statistics); and the ring search over cells smaller
than the radius (grid_ring.hip) agrees with the lane-serial kernel on the same clouds (indices and distances bit for bit, the
statistics to rounding)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_warm_search_equals_lane_serial_on_hostile_clouds(lib, seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_warm_vs_serial.py"), "60", str(seed)],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0 and "0 mismatches" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
    assert "'warm': 720" in p.stdout, p.stdout[-500:]
    # (round 6: a third context runs the ring search of grid_ring.hip wherever a finer table than the radius-sized one exists)
    assert " 0 ring mismatches" in p.stdout and "ring search" in p.stdout and ": 0 passes" not in p.stdout, p.stdout[-500:]
