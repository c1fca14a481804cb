"""BASELINE config 5: a corpus of independent orientation-constrained registrations ((scene, CAD candidate)
pairs, 24 yaw starts each, src/annotation.cpp:29-64,103-168) handed out to ranks from a shared counter.
A sample of the corpus against the oracle, and the pull queue of `bench.py --workload c5` with two ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import c5_chunk_problems, c5_corpus, c5_pick  # noqa: E402
from visma_amd import _lib, synth  # noqa: E402


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_corpus_items_equal_the_oracle_sweep(lib, oracle):
    scenes, cads, items = c5_corpus()
    assert len(items) == 192
    ctx = _lib.Context(0)
    # the two cheapest items for the CPU oracle (24 starts x 30 iterations each)
    cost = [len(cads[c]) * np.log(len(scenes[s])) for s, c in items]
    for i in np.argsort(cost)[:2]:
        s, c = items[int(i)]
        ctx.set_clouds_f64(cads[c], scenes[s])
        best, level, per = ctx.run_yaw_sweep(24, 0.05, 30)
        assert ctx.search_mode_used() == "exact"
        want = oracle.register_model_to_scene(cads[c], scenes[s], 24, 0.05, max_iter=30)
        assert level == want.best_level, i
        assert best.num_correspondences == want.k, i
        assert synth.rel_frobenius(best.transformation_, want.T) < 1e-9, i
        assert len(per) == 24 and max(p.num_correspondences for p in per) == best.num_correspondences
        # the same item as bench.py runs it: its 24 starts inside one batch
        res = ctx.run_batch(c5_chunk_problems(scenes, cads, [(s, c)], 0.05, 24), max_iter=30)
        lvl_b, best_b = c5_pick(res, 24)[0]
        assert lvl_b == level and best_b.num_correspondences == best.num_correspondences
        assert synth.rel_frobenius(best_b.transformation_, best.transformation_) < 1e-10


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_pull_the_corpus_from_one_counter(lib):
    """Every item is done exactly once per pass, whichever rank takes it (ranks share GPU 0 on a one-GPU box)."""
    env = dict(os.environ, VISMA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c5", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["config"]["items"] == 192
    assert "no collective" in d["config"]["parallelism"]
    assert d["items_done_by_all_ranks"] == 2 * 192          # two timed passes: every item exactly once per pass
    assert 0 < d["items_done_by_rank0"] < 2 * 192          # ... shared between the ranks
    assert d["registrations_per_sec"] > 0 and np.isfinite(d["value"])
