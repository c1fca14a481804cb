# C4 registration under larger grid cells (VISMA_ICP_GRID_CELL = cell edge in radii): how the pass time follows the
# number of candidates per query (the argument about a FINER grid)
mkdir -p gpurun_out/r6q
for f in 1.0 1.26 1.6 2.0; do
  VISMA_ICP_GRID_CELL=$f timeout 300 python - <<PY
import os, sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np
from visma_amd import _lib, synth
src, tgt, T_gt, r = synth.make_pair(262144, 4194304, motion="radius")
c = _lib.Context(0); c.set_clouds_f64(src, tgt); c.set_nn_mode(_lib.NN_GRID)
c.iterate(np.eye(4), r, 3)
ts = []
for _ in range(5):
    c.forget_winners(); t0 = time.perf_counter(); T, _ = c.iterate(np.eye(4), r, 20); ts.append(time.perf_counter() - t0)
c.set_profiling(1); c.get_timing(reset=True); c.forget_winners(); c.iterate(np.eye(4), r, 20); tm = c.get_timing(reset=True)
nl = max(tm["nn_launches"], 1)
print(json.dumps({"cell_in_radii": $f, "us_per_iteration_1_20": round(float(np.median(ts)) / 20 * 1e6, 2),
                  "candidates_per_query": round(tm["grid_candidates"] / nl / 262144, 2), "rows_per_query": round(tm["grid_candidates_27cell"] / nl / 262144, 2),
                  "certified": round(tm["grid_certified"] / nl / 262144, 3)}))
PY
done | tee gpurun_out/r6q/cell_c4.jsonl
