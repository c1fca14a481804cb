"""The summary line bench.py prints (the driver parses it: round 5's 22 KB line was NOT parsed -- VERDICT r5 item 1).

The line is built from canned full results (the builder's own round-4 / round-5 lines, kept under profiles/, which are
exactly what `run_c4` / `run_c3` / `run_c5` return) -- no GPU: it must stay under 4 KB, be strict JSON (no NaN /
Infinity), carry the contract's keys with `roofline` and `cpu_baseline` as numbers, and name the side file that holds
everything else."""
import glob
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

# (rounds 4-5: the line WAS the full result; round 6: the side file is)
CANNED = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[45]_bench_c[345].json")) +
                glob.glob(os.path.join(ROOT, "profiles", "r06_bench_c[345]_extras.json")))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _reject(name):
    raise ValueError("not strict JSON: %s" % name)


def _check(text):
    assert "\n" not in text and len(text.encode()) < 4096, len(text)
    line = json.loads(text, parse_constant=_reject)
    for k in REQUIRED:
        assert k in line, k
    assert isinstance(line["config"].get("workload"), str) and "model" not in line["config"]
    for v in line["config"].values():
        assert not isinstance(v, str) or len(v) <= 200
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma", "valu") and r["unit"] in ("GB/s", "TFLOP/s")
    assert isinstance(r["kernel"], str) and " " not in r["kernel"]          # the trace name, no prose
    for k in ("achieved", "peak", "frac", "avg_launch_ms"):
        assert isinstance(r[k], float) and r[k] > 0, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 * r["frac"]
    assert r["traffic"] is None or r["traffic"] > 0
    for k, v in r.items():
        assert not isinstance(v, (dict, list)), k
        assert not isinstance(v, str) or len(v) <= 48, k
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1 and len(c["sample"]) <= 200
    return line


@pytest.mark.parametrize("path", CANNED, ids=[os.path.basename(p) for p in CANNED])
def test_line_from_a_canned_result(path):
    full = json.load(open(path))
    line = _check(bench.compact_line(full, "bench_extras.json"))
    assert line["extras"] == "bench_extras.json"
    assert line["value"] == pytest.approx(full["value"], rel=1e-6)
    assert line["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-6)


def test_non_finite_numbers_never_reach_the_line(tmp_path):
    full = json.load(open(CANNED[-1]))
    full["roofline"]["traffic"] = float("nan")
    full["roofline"]["traffic_frac"] = float("inf")
    full["err_vs_T_gt"] = float("nan")
    side = tmp_path / "x" / "extras.json"
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(full, str(side))
    out = buf.getvalue()
    assert out.endswith("\n") and out.count("\n") == 1
    line = _check(out.strip())
    assert line["roofline"]["traffic"] is None and line["extras"] == str(side)
    # the side file: strict JSON too, and it holds the prose and the extra workloads the line dropped
    extras = json.loads(side.read_text(), parse_constant=_reject)
    assert extras["roofline"]["traffic"] is None and set(full) <= set(extras)


def test_a_line_that_cannot_fit_is_an_error_not_a_long_line():
    full = json.load(open(CANNED[-1]))
    full["config"] = dict(full["config"], **{k: "x" * 10000 for k in ("workload", "parallelism", "search", "nn", "solver")})
    text = bench.compact_line(full, "bench_extras.json")       # phrases are cut, the line still fits
    _check(text)
