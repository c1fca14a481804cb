"""BASELINE config 5: a corpus of independent orientation-constrained registrations ((scene, CAD candidate)
pairs, 24 yaw starts each, src/annotation.cpp:29-64,103-168) handed out to ranks from a shared counter.
A sample of the corpus against the oracle, and the pull queue of `bench.py --workload c5` with two ranks."""
import json
import os
import subprocess
import tempfile
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
    side = os.path.join(tempfile.mkdtemp(prefix="visma_bench_"), "extras.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c5", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline", "--extras-file", side], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1 and len(line[0]) < 4096
    assert json.loads(line[0])["n_gpus"] == 2
    d = json.load(open(side))                                # the full result: the side file the line names
    assert d["n_gpus"] == 2 and d["config"]["items"] == 192
    assert "no collective" in d["config"]["parallelism"]
    assert d["items_done_by_all_ranks"] == 2 * 192          # two timed passes: every item exactly once per pass
    assert 0 < d["items_done_by_rank0"] < 2 * 192          # ... shared between the ranks
    assert d["registrations_per_sec"] > 0 and np.isfinite(d["value"])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_native_work_queue_registers_every_item_once_and_like_the_sweep(lib, oracle):
    """visma_icp_run_corpus with TWO contexts (two host threads pulling from one counter; they share GPU 0 on a
    one-GPU box): every item done exactly once, by either thread; each item's choice is what a single
    orientation-constrained sweep of the library gives, and for a sampled item what the oracle's
    RegisterModelToScene restatement gives."""
    scenes, cads, items = c5_corpus()
    sub = items[:40]
    corpus = _lib.Corpus([(cads[c], scenes[s]) for s, c in sub], level=24, max_dist=0.05, max_iter=30, chunk=4)
    a, b = _lib.Context(0), _lib.Context(0)
    res = corpus.run([a, b])
    assert len(res) == len(sub)
    assert all(r[2] in (0, 1) for r in res)
    assert {r[2] for r in res} == {0, 1}                      # both threads took work
    one = _lib.Context(0)
    for i in (0, 7, 23, 39):
        s, c = sub[i]
        one.set_clouds_f64(cads[c], scenes[s])
        best, level, per = one.run_yaw_sweep(24, 0.05, 30)
        got, lvl, dev, its = res[i]
        assert lvl == level and got.num_correspondences == best.num_correspondences
        assert synth.rel_frobenius(got.transformation_, best.transformation_) < 1e-10
        assert its == sum(p.iterations for p in per)
    cost = [len(cads[c]) * np.log(len(scenes[s])) for s, c in sub]
    i = int(np.argmin(cost))
    s, c = sub[i]
    want = oracle.register_model_to_scene(cads[c], scenes[s], 24, 0.05, max_iter=30)
    assert res[i][1] == want.best_level and res[i][0].num_correspondences == want.k
    assert synth.rel_frobenius(res[i][0].transformation_, want.T) < 1e-9
    # a second pass over a shared counter that another "process" has already advanced: those items are not ours
    import ctypes
    cnt = ctypes.c_int64(16)
    res2 = corpus.run([a], ctypes.addressof(cnt))
    assert [r[2] for r in res2[:16]] == [-1] * 16 and all(r[2] == 0 for r in res2[16:])
    assert cnt.value >= len(sub)
    for x in (a, b, one):
        x.close()


def test_corpus_arguments_are_checked(lib):
    L = _lib.load()
    err = _lib.C.create_string_buffer(256)
    assert L.visma_icp_run_corpus(None, 0, None, 0, None, None, None, err, 256) == 1
    assert b"bad corpus" in err.value
