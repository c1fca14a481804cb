"""One registration of 20 iterations from the identity under SURVEY 8d's literal motion with the ring search (grid_ring.hip):
the workload of tools/ring_pmc.sh.  RING_NS / RING_NT / RING_R / RING_ITERS."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visma_amd import _lib, synth  # noqa: E402

ns = int(os.environ.get("RING_NS", "262144"))
nt = int(os.environ.get("RING_NT", "4194304"))
r = float(os.environ.get("RING_R", "0.15"))
iters = int(os.environ.get("RING_ITERS", "20"))
src, tgt, T_gt, _ = synth.make_pair(ns, nt, motion="fixed")
c = _lib.Context(0)
c.set_ring_search(1)
c.set_nn_mode(_lib.NN_GRID)
c.set_clouds_f64(src, tgt)
T, res = c.iterate(None, r, iters)
print(c.search_kernel_used(), c.ring_search(), res.fitness_, float(np.abs(T - T_gt).max()))
