"""What the PMC passes profile at C4.
VISMA_REGIME=converged (default): one cold pass + VISMA_PASSES - 1 (default 46) warm-started ones -- the last 20 are a
  host loop of their own: ONE dispatch of nn_coop_kernel_persist running 20 passes at the converged pose (bench.py's
  `value_converged`; tools/pmc_summarize.py tabulates the last dispatch separately).
VISMA_REGIME=initial: bench.py's `value` since round 5 -- after 5 warm-up iterations, three FRESH registrations of 20
  iterations from the identity (winners forgotten before each): the last dispatch of nn_coop_kernel_persist is such a
  registration's 19 warm passes (the cold pass is the lane-serial kernel's launch before it; with
  VISMA_ICP_COLD_IN_LAUNCH=1 it runs inside the launch too: 20 passes).
VISMA_PAIR=partial: the whole model against a scan of half of its surface (synth.make_partial_pair) instead of the
  full-overlap pair.  VISMA_NS overrides the source size (the saturated launches of bench.py's roofline_saturated;
  several queries per lane: one launch per pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visma_amd import _lib, synth
ns = int(os.environ.get("VISMA_NS", "262144"))
nt = 4194304
regime = os.environ.get("VISMA_REGIME", "converged")
if os.environ.get("VISMA_PAIR") == "partial":
    src, tgt, T_gt, r = synth.make_partial_pair(ns, nt, overlap=0.5)
elif ns == 262144:
    src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
else:
    _, tgt, T_gt, r = synth.make_pair(1024, nt, motion="radius")
    src = synth.make_source(ns, nt, seed_s=5678 + ns % 9973)
c = _lib.Context(0); c.set_clouds_f64(src, tgt); c.set_nn_mode(_lib.NN_GRID)
if regime == "initial":
    c.iterate(np.eye(4), r, 5)
    for _ in range(3):
        c.forget_winners()
        c.iterate(np.eye(4), r, 20)
else:
    n = int(os.environ.get('VISMA_PASSES', '47'))
    T, _ = c.iterate(np.eye(4), r, max(n - 20, 1))
    if n > 21:
        c.iterate(T, r, 20)
c.close()
