"""Time the one-off setup of a registration (upload + grid build) next to its iterations (GPU box)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visma_amd import _lib, synth

for ns, nt in ((5000, 20000), (50000, 200000), (262144, 4194304)):
    src, tgt, T, r = synth.make_pair(ns, nt, motion="radius")
    ctx = _lib.Context(0)
    ctx.set_clouds_f64(src[:100], tgt[:1000]); ctx.run(None, r, 1, 0, 0)      # warm the context
    t0 = time.perf_counter(); ctx.set_clouds_f64(src, tgt); t_first_up = time.perf_counter() - t0
    t0 = time.perf_counter(); ctx.set_clouds_f64(src, tgt); t_up = time.perf_counter() - t0      # staging already sized
    t0 = time.perf_counter(); ctx.run(None, r, 0, 0, 0); t_first = time.perf_counter() - t0   # grid build + 1 pass
    t0 = time.perf_counter(); res = ctx.run(None, r, 30, 0, 0); t_run = time.perf_counter() - t0
    print(json.dumps(dict(ns=ns, nt=nt, first_set_clouds_ms=t_first_up * 1e3, set_clouds_ms=t_up * 1e3, grid_build_plus_first_pass_ms=t_first * 1e3,
                          run_30_iterations_ms=t_run * 1e3, end_to_end_ms=(t_up + t_first + t_run) * 1e3)))
