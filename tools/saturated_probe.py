import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np
from visma_amd import _lib, synth
nt = 4194304
_, tgt, T, r = synth.make_pair(1024, nt, motion="radius")
for ns in (1048576, 2097152, 4194304, 8388608):
    src = synth.make_source(ns, nt, seed_s=77 + ns % 1000)
    c = _lib.Context(0); c.set_clouds_f64(src, tgt); c.set_nn_mode(_lib.NN_GRID)
    Tm, _ = c.iterate(np.eye(4), r, 4)
    c.set_profiling(1); c.get_timing(reset=True)
    Tm, last = c.iterate(Tm, r, 5)
    tm = c.get_timing(reset=True)
    print(ns, "nn_us %.1f" % (tm["nn_ms"] / tm["nn_launches"] * 1e3), "ns/query %.1f" % (tm["nn_ms"] / tm["nn_launches"] * 1e6 / ns), c.search_kernel_used(), last.fitness_, flush=True)
    c.close()
