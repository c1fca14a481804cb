"""Sweep the grid kernel's (lanes per query G, loads in flight U, phased) at one size (GPU box).
usage: grid_lanes_sweep.py NSxNT code [code ...]   code = G + 100*U, optionally code:max_blocks"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visma_amd import _lib, synth

ns, nt = (int(x) for x in sys.argv[1].split("x"))
src, tgt, T, r = synth.make_pair(ns, nt, motion="radius")
for code in sys.argv[2:]:
    blocks = "1024"
    if ":" in code:
        code, blocks = code.split(":")
    os.environ["VISMA_ICP_GRID_BLOCKS"] = blocks
    os.environ["VISMA_ICP_GRID_LANES"] = code
    ctx = _lib.Context(0)
    ctx.set_profiling(True)
    ctx.set_search_precision(os.environ.get("SEARCH_PRECISION", "f32"))
    ctx.set_clouds_f64(src, tgt)
    ctx.set_nn_mode(_lib.NN_GRID)
    ctx.run(None, r, 1, 0, 0)
    ctx.get_timing(reset=True)
    res = ctx.run(None, r, 30, 0, 0)
    tm = ctx.get_timing(reset=True)
    ctx.set_profiling(False)
    import time
    t0 = time.perf_counter(); ctx.run(None, r, 100, 0, 0); wall = (time.perf_counter() - t0) / 101 * 1e3
    print(json.dumps(dict(code=int(code), blocks=int(blocks), wall_ms=wall, fin_ms=tm["reduce_ms"] / max(tm["reduce_launches"], 1), nn_ms=tm["nn_ms"] / tm["nn_launches"],
                          cand=tm.get("grid_candidates", 0) / max(tm["nn_launches"], 1) / ns,
                          K=res.num_correspondences, T_hash=float(res.transformation_.sum()))), flush=True)
    del ctx
