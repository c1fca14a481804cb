"""What the PMC passes profile: one cold pass (lane-serial kernel) + VISMA_PASSES - 1 (default 46) warm-started ones at C4
-- the last 20 are the regime bench.py's `value` is timed in (tools/pmc_summarize.py tabulates them separately): they are
a host loop of their own, so with persistent launches (round 4b) they are ONE dispatch of nn_coop_kernel_persist running
20 passes, after one of 26 --
(VISMA_NS overrides the source size: the saturated launches of bench.py's roofline_saturated; several queries per lane:
one launch per pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visma_amd import _lib, synth
ns = int(os.environ.get("VISMA_NS", "262144"))
nt = 4194304
if ns == 262144:
    src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
else:
    _, tgt, T_gt, r = synth.make_pair(1024, nt, motion="radius")
    src = synth.make_source(ns, nt, seed_s=5678 + ns % 9973)
c = _lib.Context(0); c.set_clouds_f64(src, tgt); c.set_nn_mode(_lib.NN_GRID)
n = int(os.environ.get('VISMA_PASSES', '47'))
T, _ = c.iterate(np.eye(4), r, max(n - 20, 1))
if n > 21:
    c.iterate(T, r, 20)
c.close()
