#!/usr/bin/env python3
"""bench.py -- ICP iterations/s of the MI355X-native registration path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c3|c5] [--nn auto|grid|brute]

`--gpus N` with N > 1 starts its own ranks (re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`);
started under torch.distributed.run already, it reads RANK / LOCAL_RANK / WORLD_SIZE.

Workloads (BASELINE.json configs):
  c4 (default; the configuration the metric is quoted on): synthetic S-surf clouds, source
      262,144 points -> target 4,194,304 points.  One STEP = one ICP iteration = one fused
      transform + exact nearest-neighbour pass of every source point against the target, the
      Jacobian/residual reduction, the fold to the 38 statistics (same launch), one host solve,
      T <- update*T.  Clouds resident in HBM before the timed region; the radius-cell grid is built
      once per target/radius outside it, like the reference's KD-tree (build time reported).
      N > 1: the SOURCE is sharded (each rank holds the full target), every rank reduces its shard
      and ONE all-reduce of the 38 f64 accumulators per iteration sums them; total work is fixed:
      "scaling": "strong".  A weak-scaling line (source N x 262,144 -> the same 4,194,304-point target) is
      measured after the timed region and reported under "weak_scaling".
  c3: all objects of one scene in flight on ONE GPU: 12 objects x 24 yaw starts = 288 small ICPs
      (source 4k..40k points, target half of it, r = 0.02, 30 iterations).  One STEP = one
      visma_icp_run_batch over the set.  N > 1: problems dealt round-robin by size, no collective.
  c5: the corpus: every scene x every CAD candidate, each a 24-yaw orientation-constrained sweep
      (src/annotation.cpp:29-64).  Ranks PULL work items from a shared counter; no collective.
      One STEP = one pass over the corpus.

NN search: `auto` (default) = the radius-cell grid, exact mode (fp32 ranking of the candidates,
the ones inside the rounding band re-ranked in f64: the reference's correspondences);
`brute` = the LDS-tiled brute-force kernel north_star names (fp32-VALU-bound).  A few brute-force
steps are always run after the timed region and reported under `brute_force`.

Prints ONE JSON line on rank 0 (contract in the task description), with the extra objects
`roofline` (dominant kernel = NN correspondence) and `cpu_baseline` (the reference itself,
oracle/_ref, timed on this host).  c4: `value` = K iterations of a FRESH registration from T = I (the
reference's loop, Registration.cpp:167-185: first pass cold, the rest warm-started while the pose moves), timed
`--blocks` times (default 7), each between a barrier + synchronize pair, winners forgotten before each: the MEDIAN,
min / max in `blocks`; `value_converged` / `converged` = the registration carried on past its convergence (what
rounds 1-4 reported as `value`: no caller gets there); `end_to_end` (host arrays in ->
transformation out: upload, grid build, 31 passes; the PCIe-inclusive rate -- never `value`),
`roofline_saturated` (the same kernel and target with 1 M and 4 M queries per launch: the chip refilled many
times over, no single-round latency chain) and `scaling_workloads` (what N = 1, 2, 4, 8 runs of this command
can be divided by each other: C4 strong with the grid, C4 LARGE strong -- 16.8 M queries against the same target,
1.4-1.8 ms per iteration at N = 1, source-sharded --, C4 weak, C5 replicas; north_star's brute-force kernel is frozen as
a cross-check since round 4 and reported under `brute_force` only).

Ranks meet over torch.distributed's GLOO backend (handles, unique ids, barriers, MAX of the elapsed times): PyTorch's
own NCCL/RCCL backend is never initialised, the library brings up its own transport (peer-to-peer mailboxes over
xGMI, else its own RCCL communicator, else the gloo callback) -- VISMA_BENCH_BACKEND=nccl restores the old rendezvous.
"""
import argparse
import json
import math
import os
import re
import socket
import subprocess
import sys
import time

# the CPU baseline's OpenMP runtime reads this when it is loaded (all cores, threads pinned close)
os.environ.setdefault("OMP_PROC_BIND", "close")
# the host driver only supports dmabuf IPC: ranks started by an external torchrun must get this too, before the HIP
# runtime is loaded, or hipIpcGetMemHandle fails and the fastest transport (peer mailboxes over xGMI) is lost
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

NS_DEFAULT = 262144
NT_DEFAULT = 4194304
PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
PEAK_FP32_TFLOPS = 157.3        # ... FP32 vector peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["c4", "c3", "c5"], default="c4")
    ap.add_argument("--ns", type=int, default=NS_DEFAULT)
    ap.add_argument("--nt", type=int, default=NT_DEFAULT)
    ap.add_argument("--nn", choices=["auto", "grid", "brute"], default="auto")
    ap.add_argument("--shard", choices=["source", "target"], default="source",
                    help="multi-GPU decomposition of c4: source points (one all-reduce of 38 f64 per iteration; "
                         "default) or target points (north_star's wording: MIN all-reduce of NS keys + the "
                         "same sum; for targets that exceed one GPU)")
    ap.add_argument("--brute-steps", type=int, default=3)
    ap.add_argument("--f32-steps", type=int, default=10)
    ap.add_argument("--no-weak", action="store_true", help="skip the weak-scaling line of c4 at N > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=20)
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=7, help="how many times the K-step block is timed (value = median)")
    ap.add_argument("--extras-file", default=os.path.join(ROOT, "bench_extras.json"),
                    help="side file for the full result: every extra workload and note (the line on stdout carries the "
                         "contract's keys only and names this file)")
    ap.add_argument("--no-extras", action="store_true",
                    help="c4: skip from_initial_pose / end_to_end / roofline_saturated / scaling_workloads")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------
# the ONE line: numbers and one-phrase strings, <= 4 KB; everything else goes to the side file
# ------------------------------------------------------------------------------------------
LINE_LIMIT = 4096
REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_frac", "avg_launch_ms", "ms_per_pass",
                 "passes_per_launch", "launches_timed", "alg_bytes_per_launch", "examined_bytes_per_launch",
                 "alg_flops_per_launch", "frac_on_examined_bytes", "candidates_per_query", "certified_fraction",
                 "cold_pass_avg_ms")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "ms_per_iter", "gpu_vs_cpu_rel_frobenius")
CONFIG_KEYS = ("workload", "ns", "nt", "radius", "solver", "nn", "search", "problems", "work_items", "parallelism")
SCALAR_EXTRAS = ("value_converged", "value_partial_overlap", "value_literal_T_gt", "fitness", "inlier_rmse", "err_vs_T_gt",
                 "ranks_hold_identical_transforms", "matched_corr_per_sec", "problems_per_sec", "registrations_per_sec")


def _num(x):
    """numbers to 7 significant digits (the side file keeps them in full); strict JSON: no NaN / Infinity"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    x = float(x)
    if not math.isfinite(x):
        return None
    return float("%.7g" % x)


def _phrase(s, limit=160):
    s = " ".join(str(s).split())
    return s if len(s) <= limit else s[:limit - 3] + "..."


def kernel_trace_name(roofline):
    """the name rocprofv3's kernel trace shows for the launches the roofline object describes (no prose)"""
    if roofline.get("kernel_name"):
        return roofline["kernel_name"]
    k = str(roofline.get("kernel", ""))
    if roofline.get("launch", {}).get("persistent"):
        return "nn_coop_kernel_persist"
    return re.split(r"[ (]", k, 1)[0] if k else None


def compact_line(full, extras_path=None):
    """The summary line of the contract from the full result: the required keys, `config` (one short phrase per key),
    `roofline` and `cpu_baseline` as NUMBERS, a handful of scalar extras, and the path of the side file that holds the
    rest (every other workload, every note).  Always strict JSON of at most LINE_LIMIT bytes."""
    line = {k: _num(full.get(k)) for k in REQUIRED_KEYS}
    cfg = full.get("config", {})
    short = full.get("config_short", {})
    line["config"] = {k: (_phrase(short.get(k, cfg[k]), 200 if k == "workload" else 120) if isinstance(short.get(k, cfg[k]), str)
                          else _num(short.get(k, cfg[k]))) for k in CONFIG_KEYS if k in cfg or k in short}
    r = full.get("roofline")
    if r:
        rl = {"kernel": kernel_trace_name(r)}
        launch = r.get("launch", {})
        for k in ROOFLINE_KEYS:
            v = r.get(k, launch.get(k))
            if v is not None or k == "traffic":
                rl[k] = _num(v)
        if "one_pass_launch_avg_ms" in launch:
            rl["cold_pass_avg_ms"] = _num(launch["one_pass_launch_avg_ms"])
        line["roofline"] = rl
    c = full.get("cpu_baseline")
    if c:
        line["cpu_baseline"] = {k: (_phrase(c[k], 200) if isinstance(c[k], str) else _num(c[k])) for k in CPU_KEYS if k in c}
    for k in SCALAR_EXTRAS:
        if k in full and not isinstance(full[k], (dict, list)):
            line[k] = _num(full[k])
    if extras_path:
        line["extras"] = extras_path
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    # (cannot happen with the key lists above; the contract matters more than any optional key)
    for k in SCALAR_EXTRAS + ("extras",):
        if len(text) <= LINE_LIMIT:
            break
        line.pop(k, None)
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) > LINE_LIMIT:
        raise RuntimeError("bench line of %d bytes" % len(text))
    return text


def emit(full, extras_path):
    """Rank 0: the full result to the side file (indented JSON; NaN / Infinity become null), the compact line to stdout."""
    def clean(o):
        if isinstance(o, dict):
            return {str(k): clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        if isinstance(o, (np.floating, float)):
            return float(o) if math.isfinite(float(o)) else None
        if isinstance(o, np.integer):
            return int(o)
        if isinstance(o, np.ndarray):
            return clean(o.tolist())
        return o
    full = clean(full)
    shown = None
    if extras_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(extras_path)), exist_ok=True)
            with open(extras_path, "w") as f:
                json.dump(full, f, indent=1, allow_nan=False)
            shown = os.path.relpath(extras_path, ROOT) if os.path.abspath(extras_path).startswith(ROOT + os.sep) else extras_path
        except OSError as e:
            print("bench: side file %s not written (%s)" % (extras_path, e), file=sys.stderr)
    print(compact_line(full, shown), flush=True)


# ------------------------------------------------------------------------------------------
# multi-rank plumbing
# ------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` started without ranks: start them (one process per GPU)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


class Ranks:
    """torch.distributed, only where there is more than one rank."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.backend = None
        self.tdev = "cpu"
        self.force_comm = os.environ.get("VISMA_ICP_FORCE_COMM") == "1"   # exercise the collective at N = 1
        if self.world > 1 or self.force_comm:
            import torch
            import torch.distributed as dist
            # VISMA_BENCH_BACKEND=gloo: dry run of the multi-rank logic on a box with fewer GPUs than ranks
            # (gloo: the rendezvous moves a few hundred bytes; PyTorch's bundled RCCL stays out of the process's way)
            self.backend = os.environ.get("VISMA_BENCH_BACKEND", "gloo")
            self.local_rank = self.local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(self.local_rank)
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
                self.tdev = "cuda"
            else:
                import datetime
                dist.init_process_group(backend=self.backend, timeout=datetime.timedelta(
                    seconds=float(os.environ.get("VISMA_BENCH_RENDEZVOUS_TIMEOUT_S", "300"))))
            self.dist = dist
            self.torch = torch

    def barrier_sync(self):
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def reduce_max(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.tdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.tdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


BRINGUP_TIMEOUT_S = float(os.environ.get("VISMA_BENCH_BRINGUP_TIMEOUT_S", "90"))


def bounded(what, fn, *a):
    """One bring-up step of a transport (hipIpcOpenMemHandle, ncclCommInitRank, ...) under a wall-clock limit: the
    call runs on a helper thread (ctypes releases the GIL), and a call that has not returned after
    VISMA_BENCH_BRINGUP_TIMEOUT_S counts as FAILED -- the rank goes on to the agreement with its peers (every step is
    followed by a MIN over the ranks' flags) and everybody lands on the next transport together.  The stuck thread and
    its context are abandoned, not joined: a hung driver call must not eat the lease."""
    import threading
    box = {}

    def run():
        try:
            box["v"] = fn(*a)
        except BaseException as e:      # noqa: BLE001
            box["e"] = e
    t = threading.Thread(target=run, daemon=True, name="bringup:" + what)
    t.start()
    t.join(BRINGUP_TIMEOUT_S)
    if t.is_alive():
        raise TimeoutError("%s did not return within %.0f s" % (what, BRINGUP_TIMEOUT_S))
    if "e" in box:
        raise box["e"]
    return box.get("v")


def attach_comm(R, ctx, make_ctx, args):
    """Bring up the library's all-reduce on every rank together.  Preference: the library's own RCCL
    communicator (ncclAllReduce of 38 f64 on the context's stream); should it fail to come up on this
    node, every rank falls back TOGETHER to the same exchange through torch.distributed."""
    from visma_amd import _lib
    torch, dist = R.torch, R.dist
    want = os.environ.get("VISMA_BENCH_COMM", "")            # "", "ipc", "rccl", "torch"
    # 1. peer-to-peer mailboxes mapped through hipIpc: one small launch per iteration, no RCCL call
    if want in ("", "ipc") and args.shard == "source":
        ok = 1
        try:
            # (VISMA_BENCH_FAIL_IPC_EXPORT_RANK / _INIT_RANK = k: rank k's bring-up fails on purpose -- the test of the
            #  fall-back chain: every rank must land on the NEXT transport together, tests/test_multi_gpu_bench.py)
            if os.environ.get("VISMA_BENCH_FAIL_IPC_EXPORT_RANK") == str(R.rank):
                raise RuntimeError("injected failure")
            mine = bounded("hipIpcGetMemHandle", ctx.comm_ipc_export)
        except Exception as e:      # noqa: BLE001
            print("bench: rank %d: mailbox export failed (%s)" % (R.rank, e), file=sys.stderr)
            mine, ok = bytes(_lib.IPC_HANDLE_BYTES), 0
        t = torch.tensor(list(mine), dtype=torch.uint8, device=R.tdev)
        lst = [torch.zeros_like(t) for _ in range(R.world)]
        dist.all_gather(lst, t)
        flag = torch.tensor([ok], dtype=torch.int32, device=R.tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()):
            try:
                if os.environ.get("VISMA_BENCH_FAIL_IPC_INIT_RANK") == str(R.rank):
                    raise RuntimeError("injected failure")
                if os.environ.get("VISMA_BENCH_HANG_IPC_INIT_RANK") == str(R.rank):
                    bounded("hipIpcOpenMemHandle (injected hang)", time.sleep, 3600.0)      # (test: a driver call that never returns)
                bounded("hipIpcOpenMemHandle + handshake", ctx.comm_ipc_init, R.rank, R.world, [bytes(x.cpu().tolist()) for x in lst])
            except Exception as e:      # noqa: BLE001
                print("bench: rank %d: mailbox mapping failed (%s)" % (R.rank, e), file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=R.tdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()):
                return ctx, "peer-to-peer mailboxes over xGMI (hipIpc, %d ranks; one launch per iteration)" % R.world
            ctx = make_ctx()                                  # some ranks mapped, some did not: start clean
    # the library's own RCCL communicator needs one device per rank (RCCL refuses two ranks on one GPU)
    distinct = torch.cuda.device_count() >= R.world or os.environ.get("VISMA_BENCH_RCCL_SHARED_GPU") == "1"
    ok = 0 if (os.environ.get("VISMA_BENCH_FORCE_TORCH_COMM") == "1" or want == "torch" or not distinct) else 1
    uid_bytes = bytes(_lib.UNIQUE_ID_BYTES)
    if R.rank == 0 and ok:
        try:
            uid_bytes = bounded("ncclGetUniqueId", _lib.comm_unique_id)
        except Exception as e:      # noqa: BLE001
            print("bench: ncclGetUniqueId failed (%s)" % e, file=sys.stderr)
            ok = 0
    uid = torch.tensor(list(uid_bytes), dtype=torch.uint8, device=R.tdev)
    dist.broadcast(uid, 0)
    flag = torch.tensor([ok], dtype=torch.int32, device=R.tdev)
    dist.broadcast(flag, 0)
    ok = int(flag.item())
    if ok:
        try:
            bounded("ncclCommInitRank", ctx.comm_init, R.rank, R.world, bytes(uid.cpu().tolist()))
        except Exception as e:      # noqa: BLE001
            print("bench: rank %d: ncclCommInitRank failed (%s)" % (R.rank, e), file=sys.stderr)
            ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=R.tdev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()):
        return ctx, "rccl (ncclAllReduce inside the library, saw %d ranks)" % R.world
    ctx = make_ctx()                                          # a context without the half-made communicator

    def torch_allreduce(a):
        t = torch.from_numpy(a.copy()).to(R.tdev)
        dist.all_reduce(t)
        a[:] = t.cpu().numpy()

    def torch_minreduce(a):
        # element-wise MIN of unsigned 64-bit keys: flip the top bit, compare as signed, flip back
        top = np.uint64(1 << 63)
        t = torch.from_numpy((a ^ top).view(np.int64).copy()).to(R.tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        a[:] = t.cpu().numpy().view(np.uint64) ^ top
    ctx.set_allreduce(torch_allreduce, R.rank, R.world)
    if args.shard == "target":
        ctx.set_minreduce(torch_minreduce)                    # (host loop: two MIN exchanges + the sum per pass)
    return ctx, "torch.distributed callback (%s, %d ranks)" % (R.backend, R.world)


# ------------------------------------------------------------------------------------------
# CPU baseline: the reference itself where oracle/_ref is present
# ------------------------------------------------------------------------------------------
def cpu_registration():
    from oracle.oracle import Oracle, Ref
    o = Oracle()
    if Ref.available():
        r = Ref()                   # (same OpenMP runtime in this process: omp_get_max_threads() of the port's library)
        return "reference", o.num_threads(), lambda s, t, rad, m, init=None: r.registration_icp(
            s, t, rad, init=init, max_iter=m, rel_fitness=0.0, rel_rmse=0.0)
    return "port", o.num_threads(), lambda s, t, rad, m, init=None: o.registration_icp(
        s, t, rad, init=init, max_iter=m, rel_fitness=0.0, rel_rmse=0.0, grid=True)


def cpu_baseline_c4(src, tgt, radius, iters, repeats):
    """Steady-state rate = (t[1 + iters its] - t[1 it]) / iters (the one-off KD-tree build is not charged to
    the iterations), median over `repeats` pairs of runs.  Bounded sample: the SAME clouds."""
    kind, threads, run = cpu_registration()
    rates, t1s = [], []
    res = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        run(src, tgt, radius, 1)
        t1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        res = run(src, tgt, radius, 1 + iters)
        t2 = time.perf_counter() - t0
        rates.append(iters / max(t2 - t1, 1e-9))
        t1s.append(t1)
    return {
        "value": float(np.median(rates)), "unit": "ICP iterations/s", "cores": int(threads), "kind": kind,
        "sample": "same clouds %d->%d, r=%.4g: %d steady-state iterations (t[%d its]-t[1 it]), median of %d; "
                  "%d OpenMP threads of %d CPUs" % (len(src), len(tgt), radius, iters, iters + 1, repeats, int(threads),
                                                    os.cpu_count() or 0),
        "omp_proc_bind": os.environ.get("OMP_PROC_BIND", "unset"),
        "runs": [float(x) for x in rates], "ms_per_iter": 1e3 / float(np.median(rates)),
        "setup_plus_first_iter_s": float(np.median(t1s)), "T": np.asarray(res.T).tolist(),
    }


# ------------------------------------------------------------------------------------------
# rooflines
# ------------------------------------------------------------------------------------------
def load_traffic(kind, ns_local, nt, field="hbm_bytes_per_nn_launch"):
    """HBM bytes per launch from the PMC passes of a PROFILED run of this same command (profiles/traffic.json);
    not collected in this run."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return tj.get("%s:%dx%d" % (kind, ns_local, nt), {}).get(field)
    except Exception:
        return None


def batch_traffic(key):
    """fabric bytes per launch of the batch kernel of config 3 / 5 (profiles/traffic.json; PMC passes of a profiled run)"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key, {}).get("hbm_bytes_per_nn_launch")
    except Exception:
        return None


def brute_roofline(ns_local, nt, nn_ms, tile, traffic):
    # ALGORITHMIC work of ONE brute-force NN launch on one rank:
    #   flops = 8 * NS_local * NT        (3 sub, 1 mul, 2 fma = 8 flop per pair)
    #   bytes = ceil(NS_local/S_TILE) * NT * 16  +  NS_local * 24
    flops = 8.0 * ns_local * nt
    s_tile = tile["block"] * (8 if ns_local >= 65536 else 2)
    b_alg = math.ceil(ns_local / s_tile) * nt * 16.0 + ns_local * 24.0
    nn_ms = max(nn_ms, 1e-9)
    tf = flops / (nn_ms * 1e-3) / 1e12
    return {
        "kernel": "nn_brute_kernel", "bound": "valu", "achieved": tf, "peak": PEAK_FP32_TFLOPS,
        "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS, "traffic": traffic,
        "traffic_source": "profiles/traffic.json (PMC passes of a profiled run, not this run)",
        "note": "fp32 VALU roof (157.3 TF): the brute-force pair loop is VALU-bound (~1e5 flop/B) and issues "
                "no MFMA",
        "avg_launch_ms": nn_ms, "alg_flops_per_launch": flops,
        "pairs_per_launch": float(ns_local) * nt,
        "hbm_streamed": {"alg_bytes_per_launch": b_alg, "s_tile": s_tile,
                         "achieved_gbps": b_alg / (nn_ms * 1e-3) / 1e9,
                         "frac_of_8TBps": b_alg / (nn_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                         "compulsory_bytes": nt * 16.0 + ns_local * 24.0},
    }


def warm_bytes(queries, certified, rows, cand):
    """ALGORITHMIC bytes of warm-kernel launches (round 4: certificate).  Every query streams its source point (32 B) and
    its state (32 B: the previous winner's f64 point, index, LB) in.  A CERTIFIED query writes 12 B (index, d2, the new
    LB) and touches nothing else.  A SEARCHED query: 16 B per cell-table row looked up and 12 B per candidate listed
    (both counted by the kernel), 32 B x 1.05 for the f64 winner and near-ties, 32 B of state and 8 B (index, d2) out."""
    searched = queries - certified
    return queries * 64.0 + certified * 12.0 + searched * (33.6 + 32.0 + 8.0) + 16.0 * rows + 12.0 * cand


def grid_roofline(queries, nt_total, nn_ms, cand_per_launch, rows_per_launch, traffic, exact, kernel="serial",
                  certified_per_launch=0.0):
    """ALGORITHMIC (examined) bytes of ONE grid launch on one rank.  Lane-serial kernel, per query: the source point
    (32 B f64 in the exact search, 16 B fp32 otherwise) + 8 B (index, d2) out, and in the exact search 32 B x 1.05 (the
    winner and the near-ties in f64) + 32 B (the state written for the next pass); plus 16 B per cell-table row LOOKED
    UP (all 9) and 12 B (exact: packed x,y,z) or 16 B per candidate EXAMINED (counted by the kernel).  Warm kernel:
    warm_bytes() above.
    `compulsory` = every source and target point once.  The fraction on examined bytes falls when the search gets
    smarter (fewer bytes AND less time); the fraction on compulsory bytes and `traffic_frac` (fabric bytes of the
    PMC passes) do not have that defect."""
    if exact:
        per_query = 32.0 + 8.0 + 33.6 + 32.0
        cand_bytes = 12.0
    else:
        per_query, cand_bytes = 16.0 + 8.0, 16.0
    b_alg = queries * per_query + 16.0 * rows_per_launch + cand_bytes * cand_per_launch
    if exact and kernel in ("warm", "wave"):
        b_alg = warm_bytes(queries, certified_per_launch if kernel == "warm" else 0.0, rows_per_launch, cand_per_launch)
    if kernel == "ring":
        # grid_ring.hip, per query: source 32 B + state in 32 B + state out 32 B + (index, d2) 8 B + the f64 points of the
        # rounding band 33.6 B; per row looked up two table words + its entry of the visiting order (16 B); 12 B per candidate
        b_alg = queries * (32.0 + 32.0 + 32.0 + 8.0 + 33.6) + 16.0 * rows_per_launch + 12.0 * cand_per_launch
    comp = nt_total * cand_bytes + queries * (32.0 + 8.0 if exact else 24.0)
    # SURVEY 8d / BASELINE.md 3: B_alg = ceil(NS / S_TILE) NT 16 + NS 24 -- a search that reads the target once has
    # S_TILE = NS: B_min = NT 16 + NS 24.  Since round 4 `achieved` / `frac` are quoted on THESE bytes (the same formula
    # every round; r3: 0.29): the examined bytes fell 3.3x with the certificates, and a fraction on them reports a
    # faster kernel as a worse one (r3 0.37 -> r4 0.13); it stays in the line as frac_on_examined_bytes.
    b_survey = nt_total * 16.0 + queries * 24.0
    nn_ms = max(nn_ms, 1e-9)
    gbps_examined = b_alg / (nn_ms * 1e-3) / 1e9
    gbps = b_survey / (nn_ms * 1e-3) / 1e9
    return {
        "kernel": {"warm": "nn_coop_kernel (warm-started exact search: certified queries -- winner provably unchanged -- skip "
                           "the search; the others, compacted over the workgroup: previous winner bounds the query, reachable "
                           "cells listed, one chunk list per workgroup ranked by all its waves; f64 re-rank; fold fused)",
                   "wave": "nn_wave_kernel (grid_wave.hip: round 3's warm-started wave-cooperative exact search, no certificates: "
                           "what batches and sweeps run after their first pass; fold fused)",
                   "ring": "nn_ring_kernel (grid_ring.hip: cells smaller than the radius, rows visited nearest first and bounded "
                           "by the best so far / the previous winner; fp32 ranking + f64 re-rank of the rounding band; fold fused)",
                   "serial": "nn_grid_reduce_kernel" + (" (exact: fp32 ranking + f64 re-rank of the rounding band; fold fused)"
                                                        if exact else "")}[kernel],
        "bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS,
        "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS, "traffic": traffic,
        "traffic_source": "profiles/traffic.json (PMC passes of a profiled run of this command, not this run)",
        "traffic_frac": (traffic / (nn_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS) if traffic else None,
        "avg_launch_ms": nn_ms, "alg_bytes_per_launch": b_survey,
        "alg_bytes_formula": "SURVEY 8d B_min = NT x 16 + NS x 24",
        "examined_bytes_per_launch": b_alg, "achieved_on_examined_bytes": gbps_examined,
        "frac_on_examined_bytes": gbps_examined / PEAK_HBM_GBPS,
        "candidates_per_query": cand_per_launch / max(queries, 1),
        "cell_table_rows_per_query": rows_per_launch / max(queries, 1),
        "certified_fraction": certified_per_launch / max(queries, 1),
        "compulsory_bytes": comp,
        "frac_on_compulsory_bytes": comp / (nn_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
        "note": "one launch per iteration: transform + exact NN + Jacobian/residual reduction + fold.  `frac` is on "
                "SURVEY 8d's bytes (every target point 16 B, every source point 24 B, once); `frac_on_examined_bytes` on "
                "what the search really touches (counted by the kernel: it FALLS when pruning and the certificates "
                "improve -- fewer bytes and less time); `traffic_frac` on the fabric bytes the PMC counters saw.  At "
                "262,144 queries the chip holds every wave at once and the launch lasts as long as one workgroup's chain "
                "of dependent phases: latency-bound, 70 % of the wave cycles waiting (DESIGN.md 4.1d); "
                "`roofline_saturated` shows the same kernel with the chip refilled.",
    }


# ------------------------------------------------------------------------------------------
# workload c4
# ------------------------------------------------------------------------------------------
def c4_context(R, args, src, tgt, ns, nt, prec="exact"):
    from visma_amd import _lib
    ctx = _lib.Context(R.local_rank)
    ctx.set_search_precision(prec)
    sharded = not (R.world == 1 and not R.force_comm)
    if args.shard == "source" or not sharded:
        lo, hi = (ns * R.rank) // R.world, (ns * (R.rank + 1)) // R.world
        ctx.set_clouds_f64(src[lo:hi], tgt)       # every rank centres on the (full) target centroid
        ctx.set_global_source_count(ns)
        return ctx, hi - lo, nt
    lo, hi = (nt * R.rank) // R.world, (nt * (R.rank + 1)) // R.world
    ctx.set_target_shard(lo, nt, tgt.mean(0))
    ctx.set_clouds_f64(src, tgt[lo:hi])
    return ctx, ns, hi - lo


def timed_block(R, ctx, T, radius, steps):
    """K ICP iterations continuing from T, between two barrier + synchronize pairs; the MAX over ranks."""
    R.barrier_sync()
    t0 = time.perf_counter()
    T, last = ctx.iterate(T, radius, steps)          # every step ends with the statistics on the host
    R.barrier_sync()
    return T, last, R.reduce_max(time.perf_counter() - t0)


def timed_registrations(R, ctx, radius, warmup, steps, nn_mode, prof_every, blocks=5):
    """SURVEY 8d's registration, `blocks` times: W untimed iterations (grid build, buffers, code objects), then every
    block a FRESH registration -- nothing remembered from earlier passes -- of K timed iterations from T = I: the first
    pass cold, the rest warm-started while the pose moves (Registration.cpp:167-185 under criteria (0, 0, K)).  Each
    block between two barrier + synchronize pairs.  Returns the final transform of the last block and every elapsed time."""
    from visma_amd import _lib
    ctx.set_nn_mode({"auto": _lib.NN_AUTO, "grid": _lib.NN_GRID, "brute": _lib.NN_BRUTE}[nn_mode])
    ctx.set_profiling(prof_every)
    ctx.get_timing(reset=True)
    if warmup > 0:
        ctx.iterate(np.eye(4), radius, warmup)      # also builds the grid (one-off)
    setup = ctx.get_timing(reset=True)
    elapsed, last, T = [], None, np.eye(4)
    for _ in range(max(blocks, 1)):
        ctx.forget_winners()
        ctx.get_timing()                            # (drains the stream: the reset of the winners is not in the block)
        T, last, el = timed_block(R, ctx, np.eye(4), radius, steps)
        elapsed.append(el)
    tm = ctx.get_timing(reset=True)
    if tm["nn_launches"] == 0:
        ctx.set_profiling(1)
        ctx.forget_winners()
        ctx.iterate(np.eye(4), radius, 3)
        tm = ctx.get_timing(reset=True)
    ctx.set_profiling(0)
    return T, last, elapsed, tm, setup


def timed_iterations(R, ctx, radius, warmup, steps, nn_mode, prof_every, blocks=1, T0=None):
    """W untimed iterations, then `blocks` blocks of K timed iterations each (the pose carries on from block to
    block; T0: carry on from there).  Returns the elapsed time of every block."""
    from visma_amd import _lib
    ctx.set_nn_mode({"auto": _lib.NN_AUTO, "grid": _lib.NN_GRID, "brute": _lib.NN_BRUTE}[nn_mode])
    ctx.set_profiling(prof_every)
    T = np.eye(4) if T0 is None else T0
    ctx.get_timing(reset=True)
    if warmup > 0:
        T, _ = ctx.iterate(T, radius, warmup)       # also builds the grid (one-off)
    setup = ctx.get_timing(reset=True)
    elapsed, last = [], None
    for _ in range(max(blocks, 1)):
        T, last, el = timed_block(R, ctx, T, radius, steps)
        elapsed.append(el)
    tm = ctx.get_timing(reset=True)
    if tm["nn_launches"] == 0:
        # too few steps for the sparse event timing to have seen a launch: time three more passes (outside
        # the timed region) so that the roofline object still carries a measured kernel duration
        ctx.set_profiling(1)
        ctx.iterate(T, radius, 3)
        tm = ctx.get_timing(reset=True)
    ctx.set_profiling(0)
    return T, last, elapsed, tm, setup


def persistent_launches(roofline, tm, queries=0, nt=0, traffic_key="grid_persist"):
    """Persistent launches (round 4b: ONE launch of the certificate kernel runs the passes of a host loop, the next
    transform handed over through mapped host memory): the timing counters hold whole launches -- the waits for the host
    included -- and count their passes.  `achieved` = bytes per launch / launch duration is the same ratio either way;
    the object says per LAUNCH what rocprofv3 sees (one long dispatch) and per PASS what an iteration costs."""
    pl, pp = tm.get("persist_launches", 0.0), tm.get("persist_passes", 0.0)
    if pl <= 0 or pp <= 0:
        return roofline
    per_pass_ms = roofline["avg_launch_ms"]
    k = pp / pl
    roofline["launch"] = {"persistent": True, "launches_timed": pl, "passes_timed": pp, "passes_per_launch": k,
                          "avg_launch_ms": tm["persist_ms"] / pl, "ms_per_pass": tm["persist_ms"] / pp,
                          "one_pass_launches_in_the_same_counters": tm["nn_launches"] - pp,
                          "note": "the kernel-trace name is nn_coop_kernel_persist; a launch lasts as long as its loop: "
                                  "its waits for the host's next transform (statistics out, solve, command back over "
                                  "PCIe) are inside"}
    # The object describes the PERSISTENT launches (the dominant kernel): their duration, their passes.  One-pass launches
    # the same counters hold -- the lane-serial first pass of every fresh registration -- are reported beside them.
    others = tm["nn_launches"] - pp
    per_pass_ms = tm["persist_ms"] / pp
    if others > 0:
        roofline["launch"]["one_pass_launch_avg_ms"] = (tm["nn_ms"] - tm["persist_ms"]) / others
        roofline["launch"]["one_pass_launches_are"] = "the first (cold) pass of a registration: nn_grid_reduce_kernel, lane-serial"
        # (achieved / frac were formed with the mean over ALL timed passes: re-form them for the persistent launches)
        scale = roofline["avg_launch_ms"] / per_pass_ms
        for key in ("achieved", "frac", "achieved_on_examined_bytes", "frac_on_examined_bytes", "frac_on_compulsory_bytes"):
            if roofline.get(key) is not None:
                roofline[key] = roofline[key] * scale
    per_pass_traffic = load_traffic(traffic_key, int(queries), int(nt), "hbm_bytes_per_pass") if queries else None
    if per_pass_traffic:
        roofline["traffic"] = per_pass_traffic * k
        roofline["traffic_per_pass"] = per_pass_traffic
        roofline["traffic_frac"] = per_pass_traffic / (per_pass_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS
    elif roofline.get("traffic"):
        roofline["traffic_per_pass"] = roofline["traffic"]      # (the one-pass kernel's PMC figure)
        roofline["traffic_frac"] = roofline["traffic"] / (per_pass_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS
        roofline["traffic"] = roofline["traffic"] * k
    roofline["avg_launch_ms"] = tm["persist_ms"] / pl
    for key in ("alg_bytes_per_launch", "examined_bytes_per_launch"):
        roofline[key.replace("_per_launch", "_per_pass")] = roofline[key]
        roofline[key] = roofline[key] * k
    roofline["ms_per_pass"] = per_pass_ms
    return roofline


def kernel_roofline(ctx, ns_local, nt_local, tm, traffic_kind=None, persist_traffic_key="grid_persist"):
    """roofline object of the search kernel the context's last passes ran, from its event / candidate counters"""
    nl = max(tm["nn_launches"], 1)
    kind = ctx.search_kernel_used()
    key = {"warm": "grid_warm", "serial": "grid"}.get(kind, "grid")
    r = grid_roofline(ns_local, nt_local, tm["nn_ms"] / nl, tm["grid_candidates"] / nl,
                      tm["grid_candidates_27cell"] / nl, load_traffic(traffic_kind or key, ns_local, nt_local),
                      ctx.search_mode_used() != "f32", kind if kind in ("warm", "serial", "ring") else "serial",
                      tm["grid_certified"] / nl)
    return persistent_launches(r, tm, ns_local if traffic_kind != "none" else 0, nt_local, persist_traffic_key)


def c4_variant(R, device, src, tgt, radius, T_gt, steps, what, traffic_key=None):
    """One more C4-shaped registration measured like the headline, on its own context (rank 0, N = 1 only): iterations
    1..K from the identity (first pass cold, the rest warm-started while the pose moves: the `value` regime, with the
    roofline object of ITS launches -- traffic_key: its PMC entry in profiles/traffic.json) and the K iterations after 2K
    more (converged), each the median of three."""
    from visma_amd import _lib, synth
    c = _lib.Context(device)
    c.set_clouds_f64(src, tgt)
    c.set_nn_mode(_lib.NN_GRID)
    c.iterate(np.eye(4), radius, 2)                                  # grid build, buffers
    first, cont = [], []
    T = np.eye(4)
    c.set_profiling(4)
    c.get_timing(reset=True)
    for _ in range(3):
        c.forget_winners()
        c.get_timing()
        T, last, el = timed_block(R, c, np.eye(4), radius, steps)
        first.append(el)
    tm_first = c.get_timing(reset=True)
    c.set_profiling(0)
    first_roofline = kernel_roofline(c, len(src), len(tgt), tm_first, traffic_kind="none" if traffic_key is None else "grid_initial",
                                     persist_traffic_key=traffic_key or "grid_persist")
    first_roofline.pop("note", None)
    T, _ = c.iterate(T, radius, steps)
    for _ in range(3):
        T, last, el = timed_block(R, c, T, radius, steps)
        cont.append(el)
    c.set_profiling(1)
    c.get_timing(reset=True)
    T, last = c.iterate(T, radius, 8)
    tm = c.get_timing(reset=True)
    c.set_profiling(0)
    out = {"workload": what, "ns": len(src), "nt": len(tgt), "radius": radius, "steps": steps,
           "from_initial_pose": {"icp_iterations_per_sec": steps / float(np.median(first)),
                                 "ms_per_step": float(np.median(first)) / steps * 1e3, "roofline": first_roofline},
           "continuing": {"icp_iterations_per_sec": steps / float(np.median(cont)),
                          "ms_per_step": float(np.median(cont)) / steps * 1e3,
                          "iterations": "%d..%d" % (2 * steps + 1, 5 * steps)},
           "fitness": last.fitness_, "K": last.num_correspondences, "inlier_rmse": last.inlier_rmse_,
           "err_vs_T_gt": synth.rel_frobenius(T, T_gt),
           "roofline": kernel_roofline(c, len(src), len(tgt), tm, traffic_kind="none")}
    out["roofline"]["launches_timed"] = tm["nn_launches"]
    c.close()
    return out


def c4_end_to_end(device, src, tgt, radius, iters=30, repeats=5):
    """What `RegistrationICP(source, target, ...)` costs its caller (src/evaluation.cpp:248-271): host arrays
    in, transformation out, on a context whose buffers exist (the second registration of a process)."""
    from visma_amd import _lib
    c = _lib.Context(device)
    c.set_clouds_f64(src, tgt)
    c.run(None, radius, iters, 0.0, 0.0)
    up, run = [], []
    for _ in range(repeats):
        t0 = time.perf_counter()
        c.set_clouds_f64(src, tgt)
        t1 = time.perf_counter()
        res = c.run(None, radius, iters, 0.0, 0.0)
        t2 = time.perf_counter()
        up.append(t1 - t0)
        run.append(t2 - t1)
    c.close()
    u, r = float(np.median(up)), float(np.median(run))
    return {"ns": len(src), "nt": len(tgt), "iterations": iters, "nn_passes": iters + 1,
            "upload_ms": u * 1e3, "grid_build_and_loop_ms": r * 1e3, "total_ms": (u + r) * 1e3,
            "pcie_inclusive_iterations_per_sec": iters / (u + r), "median_of": repeats,
            "fitness": res.fitness_, "K": res.num_correspondences}


def yaw_sweep_end_to_end(device, src, tgt, radius, level=24, iters=30, repeats=5):
    """What feh::RegisterModelToScene costs its caller (src/annotation.cpp:29-64): host arrays in, the best of `level`
    yaw starts out -- both clouds up once, the `level` registrations advance together (one launch per pass for all)."""
    from visma_amd import _lib
    c = _lib.Context(device)
    c.set_clouds_f64(src, tgt)
    c.run_yaw_sweep(level, radius, iters, 1e-6, 1e-6)
    up, run = [], []
    its = 0
    for _ in range(repeats):
        t0 = time.perf_counter()
        c.set_clouds_f64(src, tgt)
        t1 = time.perf_counter()
        best, which, per = c.run_yaw_sweep(level, radius, iters, 1e-6, 1e-6)
        t2 = time.perf_counter()
        up.append(t1 - t0)
        run.append(t2 - t1)
        its = sum(p.iterations for p in per)
    c.close()
    u, r = float(np.median(up)), float(np.median(run))
    return {"ns": len(src), "nt": len(tgt), "yaw_starts": level, "max_iterations": iters, "iterations_run": its,
            "upload_ms": u * 1e3, "sweep_ms": r * 1e3, "total_ms": (u + r) * 1e3, "best_start": which,
            "best_fitness": best.fitness_, "median_of": repeats,
            "pcie_inclusive_iterations_per_sec": its / (u + r)}


def c4_saturated(device, tgt, nt, radius, ns_list=(1048576, 4194304), steps=5):
    """The kernel of the timed region with the chip refilled many times: same 4 M-point target and radius,
    1 M and 4 M queries per launch (measured after the timed region, HIP events on the context's stream)."""
    from visma_amd import _lib, synth
    out = []
    for ns_s in ns_list:
        src_s = synth.make_source(ns_s, nt, seed_s=5678 + ns_s % 9973)
        c = _lib.Context(device)
        c.set_clouds_f64(src_s, tgt)
        c.set_nn_mode(_lib.NN_GRID)
        T, _ = c.iterate(np.eye(4), radius, 4)
        c.set_profiling(1)
        c.get_timing(reset=True)
        t0 = time.perf_counter()
        T, last = c.iterate(T, radius, steps)
        dt = time.perf_counter() - t0
        tm = c.get_timing(reset=True)
        c.set_profiling(0)
        r = kernel_roofline(c, ns_s, nt, tm)
        r.update(ns=ns_s, nt=nt, steps=steps, ms_per_step=dt / steps * 1e3, launches_timed=tm["nn_launches"],
                 query_iterations_per_sec=float(ns_s) * steps / dt, fitness=last.fitness_)
        r.pop("note", None)
        out.append(r)
        c.close()
    return out


def run_c4(R, args):
    from visma_amd import _lib, synth
    ns, nt = args.ns, args.nt
    src, tgt, T_gt, radius = synth.make_pair(ns, nt, motion="radius")
    ctx, ns_local, nt_local = c4_context(R, args, src, tgt, ns, nt)
    comm_kind = "none"
    if R.dist is not None:
        ctx, comm_kind = attach_comm(R, ctx, lambda: c4_context(R, args, src, tgt, ns, nt)[0], args)
    # HIP-event timing of the kernels: every launch with brute force (117 ms each), every 4th ICP pass with
    # the grid (two event records cost ~7 us of a ~50 us iteration)
    # (with fewer than 8 steps nothing is timed inside the timed region: timed_iterations measures the
    #  kernel on three extra passes afterwards)
    prof_every = 1 if args.nn == "brute" else (4 if args.steps >= 8 else 0)
    blocks = 1 if args.nn == "brute" else max(args.blocks, 1)
    # `value` (since round 5): SURVEY 8d's registration as the reference runs it -- K iterations from T = I, the cold
    # first pass included (Registration.cpp:167-185) --, the median of `blocks` fresh registrations
    T, last, elapsed_all, tm, setup = timed_registrations(R, ctx, radius, args.warmup, args.steps, args.nn, prof_every, blocks)
    elapsed = float(np.median(elapsed_all))
    mode = "grid" if ctx.nn_mode_used() == _lib.NN_GRID else "brute"
    search = ctx.search_mode_used()
    kernel_kind = ctx.search_kernel_used()
    # every rank solved from the same all-reduced statistics: their transforms must agree to the bit
    ranks_agree = True
    if R.dist is not None:
        tt = R.torch.tensor(np.asarray(T, np.float64).ravel(), dtype=R.torch.float64, device=R.tdev)
        hi, lo = tt.clone(), tt.clone()
        R.dist.all_reduce(hi, op=R.dist.ReduceOp.MAX)
        R.dist.all_reduce(lo, op=R.dist.ReduceOp.MIN)
        ranks_agree = bool(R.torch.equal(hi, lo))
    nl = max(tm["nn_launches"], 1)
    nn_ms = R.reduce_max(tm["nn_ms"] / nl)
    cand = tm["grid_candidates"] / nl

    # the regime rounds 1-4 reported as `value`: the registration carried on past its convergence (2K more iterations
    # untimed, then `blocks` blocks of K continuing at the converged pose: nearly every query certified) -- no caller of
    # the reference gets there (its default criteria stop a C4 registration at iteration 29); kept as `value_converged`
    converged = None
    if mode == "grid" and not args.no_extras:
        Tcv, lastc, el_c, tm_c, _ = timed_iterations(R, ctx, radius, 2 * args.steps, args.steps, args.nn, prof_every, blocks, T0=T)
        e_c = float(np.median(el_c))
        converged = {"steps": args.steps, "ms_per_step": e_c / args.steps * 1e3, "icp_iterations_per_sec": args.steps / e_c,
                     "blocks": [args.steps / e for e in el_c], "median_of": len(el_c),
                     "iterations": "%d..%d of one registration" % (3 * args.steps + 1, (3 + len(el_c)) * args.steps),
                     "tm": tm_c, "fitness": lastc.fitness_, "err_vs_T_gt": synth.rel_frobenius(Tcv, T_gt),
                     # (collective: every rank is here)
                     "nn_ms": R.reduce_max(tm_c["nn_ms"] / max(tm_c["nn_launches"], 1))}

    # north_star's brute-force kernel on the same (sharded) problem, outside the timed region: collective timing
    brute = None
    if mode != "brute" and args.brute_steps > 0:
        ctx.set_nn_mode(_lib.NN_BRUTE)
        ctx.set_profiling(1)
        ctx.iterate(T, radius, 1)
        ctx.get_timing(reset=True)
        Tb, _, tb = timed_block(R, ctx, np.eye(4), radius, args.brute_steps)
        tmb = ctx.get_timing(reset=True)
        ctx.set_profiling(0)
        Tg, _ = (ctx.set_nn_mode(_lib.NN_GRID), ctx.iterate(np.eye(4), radius, args.brute_steps))[1]
        brute = {"steps": args.brute_steps, "ms_per_step": tb / args.brute_steps * 1e3,
                 "nn_ms": R.reduce_max(tmb["nn_ms"] / max(tmb["nn_launches"], 1)),
                 "rel_frobenius_vs_grid": synth.rel_frobenius(Tb, Tg)}
        ctx.set_nn_mode({"auto": _lib.NN_AUTO, "grid": _lib.NN_GRID}[args.nn])

    # the same workload with the plain fp32 search of round 1 (no f64 re-rank), outside the timed region
    f32_extra = None
    if R.world == 1 and mode == "grid" and search == "exact" and args.f32_steps > 0:
        c32 = _lib.Context(R.local_rank)
        c32.set_search_precision("f32")
        c32.set_clouds_f64(src, tgt)
        c32.set_nn_mode(_lib.NN_GRID)
        c32.iterate(np.eye(4), radius, 2)
        t0f = time.perf_counter()
        T32, _ = c32.iterate(np.eye(4), radius, args.f32_steps)
        tf = time.perf_counter() - t0f
        Tex, _ = ctx.iterate(np.eye(4), radius, args.f32_steps)
        f32_extra = {"steps": args.f32_steps, "iterations_per_sec": args.f32_steps / tf,
                     "rel_frobenius_vs_exact_search": synth.rel_frobenius(T32, Tex),
                     "note": "fp32 ranking only (round-1 kernel, lane-serial): may decide near-ties / radius cases "
                             "differently from the reference"}
        c32.close()

    # weak scaling (after the timed region): N x 262,144 source points against the SAME 4,194,304-point target
    # at the same radius -- every rank's launch is the N = 1 launch, only the exchange is added
    weak = None
    if R.world > 1 and not args.no_weak and args.shard == "source":
        wns, wnt = ns * R.world, nt
        wsrc = synth.make_source(wns, wnt)
        wctx, wns_local, _ = c4_context(R, args, wsrc, tgt, wns, wnt)
        wctx, _ = attach_comm(R, wctx, lambda: c4_context(R, args, wsrc, tgt, wns, wnt)[0], args)
        _, wlast, wel, wtm, _ = timed_iterations(R, wctx, radius, args.warmup, args.steps, args.nn, 4, 3)
        welapsed = float(np.median(wel))
        weak = {"ns": wns, "nt": wnt, "radius": radius, "ms_per_step": welapsed / args.steps * 1e3,
                "icp_iterations_per_sec": args.steps / welapsed,
                "point_iterations_per_sec": float(wns) * args.steps / welapsed,
                "nn_kernel_ms": R.reduce_max(wtm["nn_ms"] / max(wtm["nn_launches"], 1)),
                "fitness": wlast.fitness_, "median_of_blocks": 3,
                "note": "per-rank work fixed (262,144 queries per rank against the same target and radius as at N = 1): "
                        "ideal weak scaling keeps icp_iterations_per_sec at the N = 1 value of this line's `value`"}
        wctx.close()

    # a LARGE source against the same target, source-sharded (strong scaling): 64 x the headline's queries -- the one grid
    # workload whose iteration is long enough (1 ms at N = 1) for eight ranks to divide it
    large = None
    if not args.no_extras and mode == "grid" and args.shard == "source" and args.nn != "brute":
        lns = 64 * ns
        lsrc = synth.make_source(lns, nt, seed_s=97)
        lctx, lns_local, _ = c4_context(R, args, lsrc, tgt, lns, nt)
        if R.dist is not None:
            lctx, _ = attach_comm(R, lctx, lambda: c4_context(R, args, lsrc, tgt, lns, nt)[0], args)
        _, llast, lel, _, _ = timed_iterations(R, lctx, radius, 2, 10, args.nn, 0, 3)
        lelapsed = float(np.median(lel))
        large = {"ns": lns, "nt": nt, "queries_per_rank": lns_local, "steps": 10, "ms_per_step": lelapsed / 10 * 1e3,
                 "icp_iterations_per_sec": 10 / lelapsed, "fitness": llast.fitness_, "median_of_blocks": 3,
                 "expectation": "per-rank launches of 16.8 M / 8.4 M / 4.2 M / 2.1 M queries at N = 1 / 2 / 4 / 8 against "
                                "the full target; measured on one GPU at those sizes (tools/ab_probe.py, "
                                "profiles/r04_ab_probe.jsonl): 1.4-1.8 ms / 0.45-0.60 ms / 0.24-0.31 ms / 0.12-0.17 ms "
                                "per iteration -- 8x or more at 8 GPUs before the 608-byte exchange costs anything (above "
                                "8.4 M queries a lane takes two queries in turn: the one-GPU launch is the slow one)"}
        lctx.close()
        del lsrc

    # configuration 5 (replicas only) for a few passes, so that the N = 1, 2, 4, 8 lines of this same command
    # also carry a workload that shards without any exchange
    c5 = None
    if not args.no_extras and args.nn != "brute":
        sub = argparse.Namespace(**vars(args))
        sub.steps, sub.warmup, sub.no_cpu_baseline = 2, 1, True
        c5r = run_c5(R, sub, tag="sw")
        if c5r is not None:
            c5 = {"icp_iterations_per_sec": c5r["value"], "registrations_per_sec": c5r["registrations_per_sec"],
                  "ms_per_pass": c5r["ms_per_step"], "passes": 2, "parallelism": c5r["config"]["parallelism"]}

    out = None
    if R.rank == 0:
        tile = _lib.tile_config()
        exact = search != "f32"
        if mode == "grid":
            # (traffic: PMC passes of tools/run_c4_iterations.py in ITS regime -- fresh registrations from the identity;
            #  the converged regime's counters belong to `converged.roofline`)
            roofline = grid_roofline(ns_local, nt_local, nn_ms, cand, tm["grid_candidates_27cell"] / nl,
                                     None,
                                     exact, kernel_kind if kernel_kind in ("warm", "serial") else "serial",
                                     tm["grid_certified"] / nl)
            roofline = persistent_launches(roofline, tm, ns_local, nt_local, "grid_persist_initial")
        else:
            roofline = brute_roofline(ns_local, nt_local, nn_ms, tile, load_traffic("brute", ns_local, nt_local))
        roofline["launches_timed"] = tm["nn_launches"] if "launch" not in roofline else roofline["launch"]["launches_timed"]
        roofline["timed_every_nth_pass"] = prof_every if "launch" not in roofline else 1
        roofline["separate_fold_launch_avg_ms"] = tm["reduce_ms"] / max(tm["reduce_launches"], 1)
        roofline["regime"] = "the timed region: %d fresh registrations of %d iterations from T = I (cold pass + warm passes)" % (
            len(elapsed_all), args.steps)
        par = ("source-sharded x%d, 1 all-reduce(38 f64)/iter via %s" % (R.world, comm_kind)
               if args.shard == "source" else
               "target-sharded x%d, ncclAllReduce(min, %d u64) + ncclAllReduce(38 f64)/iter via %s" % (R.world, ns, comm_kind))
        launch_mode = ("ONE persistent launch per host loop (passes inside it)" if tm.get("persist_passes", 0) > 0
                       else "one launch per pass")
        if R.dist is not None:
            par += "; launch mode: %s%s; ranks meet over torch.distributed/%s" % (
                launch_mode, " (VISMA_ICP_PERSIST_RANKS=1)" if os.environ.get("VISMA_ICP_PERSIST_RANKS") == "1" else
                " (default for ranks; VISMA_ICP_PERSIST_RANKS=1 keeps the ranks' launches alive across passes)", R.backend)
        else:
            par += "; launch mode: %s" % launch_mode
        out = {
            "metric": "icp_iterations_per_sec", "value": args.steps / elapsed,
            "unit": "ICP iterations/s", "n_gpus": R.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C4 S-surf %d-pt source -> %d-pt target, %d fixed ICP iterations from T = I (a fresh "
                                   "registration: cold first pass + warm passes; median of %d), nn=%s, search=%s" % (
                                       ns, nt, args.steps, len(elapsed_all), mode, search),
                       "ns": ns, "nt": nt, "radius": radius, "solver": "kabsch", "nn": mode, "search": search,
                       "search_kernel": kernel_kind,
                       "arithmetic": "fp32 candidate ranking, f64 re-rank of the rounding band, f64 statistics",
                       "parallelism": par},
            "blocks": {"timed": len(elapsed_all), "steps_each": args.steps,
                       "value_is": "median over fresh registrations of K iterations from the identity (SURVEY 8d; "
                                   "Registration.cpp:167-185), winners forgotten before each",
                       "iterations_per_sec": [args.steps / e for e in elapsed_all],
                       "min": args.steps / max(elapsed_all), "max": args.steps / min(elapsed_all)},
            "candidates_evaluated_per_sec": cand * R.world * args.steps / elapsed if mode == "grid" else float(ns) * nt * args.steps / elapsed,
            "equivalent_bruteforce_mpairs_per_sec": float(ns) * nt * args.steps / elapsed / 1e6,
            "matched_corr_per_sec": last.num_correspondences * args.steps / elapsed,
            "fitness": last.fitness_, "inlier_rmse": last.inlier_rmse_, "ranks_hold_identical_transforms": ranks_agree,
            "err_vs_T_gt": synth.rel_frobenius(T, T_gt),
            "setup_ms": {"grid_build_kernels": setup["aux_ms"]},
            "roofline": roofline,
        }
        # one phrase per key for the line on stdout (the long forms above stay in the side file)
        out["config_short"] = {
            "workload": "C4 S-surf %d->%d, %d ICP iterations of a fresh registration from T=I (cold pass + warm passes), "
                        "median of %d" % (ns, nt, args.steps, len(elapsed_all)),
            "parallelism": ("1 GPU, %s" % launch_mode.split(" (")[0] if R.dist is None else
                            "%s-sharded x%d, allreduce(38 f64)/iter via %s" % (args.shard, R.world, comm_kind.split(" (")[0]))}
        # (continuity with rounds 3-4, where this regime was an extra key and `value` the converged one)
        out["value_from_initial_pose"] = out["value"]
        if converged is not None:
            tm_c = converged.pop("tm")
            nlc = max(tm_c["nn_launches"], 1)
            rc = grid_roofline(ns_local, nt_local, converged.pop("nn_ms"), tm_c["grid_candidates"] / nlc,
                               tm_c["grid_candidates_27cell"] / nlc, load_traffic("grid_warm", ns_local, nt_local), exact,
                               kernel_kind if kernel_kind in ("warm", "serial") else "serial", tm_c["grid_certified"] / nlc)
            rc = persistent_launches(rc, tm_c, ns_local, nt_local)
            rc.pop("note", None)
            converged["roofline"] = rc
            out["converged"] = converged
            out["value_converged"] = converged["icp_iterations_per_sec"]
            out["config"]["workload"] += "; value_converged = iterations %s, continuing at the converged pose: %.0f it/s " \
                                         "(what rounds 1-4 reported as value)" % (converged["iterations"],
                                                                                converged["icp_iterations_per_sec"])
        if R.world == 1 and not args.no_extras and mode == "grid":
            # what the reference's callers register: a model against a PARTIAL scan (fitness ~0.5: half of the queries
            # find nothing within the radius, every pass), same sizes, same radius rule, same motion
            psrc, ptgt, pT, pr = synth.make_partial_pair(ns, nt, overlap=0.5)
            out["partial_overlap"] = c4_variant(R, R.local_rank, psrc, ptgt, pr, pT, args.steps,
                                                "C4 sizes, the whole model against a scan of half of its surface "
                                                "(synth.make_partial_pair; the compiled reference's result on it: "
                                                "tests/golden/c4_partial_ref.npz)",
                                                traffic_key="grid_persist_partial_initial" if (ns, nt) == (NS_DEFAULT, NT_DEFAULT) else None)
            out["value_partial_overlap"] = out["partial_overlap"]["from_initial_pose"]["icp_iterations_per_sec"]
            out["value_partial_overlap_converged"] = out["partial_overlap"]["continuing"]["icp_iterations_per_sec"]
            out["value_partial_overlap_from_initial_pose"] = out["value_partial_overlap"]
            del psrc, ptgt
            # SURVEY 8d's literal ground truth (5 deg yaw, 1 deg pitch, ~3 cm) needs a radius of 0.15 m to converge: at
            # 4,194,304 target points that radius holds 15,000 points per query and 3,200 per occupied radius-sized cell --
            # the regime of the ring search over cells of a few point spacings (grid_ring.hip, round 6; chosen by the
            # library from the occupancy).  At the headline's sizes, and at 16,384 -> 65,536 (52 points per cell)
            lsrc, ltgt, lT, _ = synth.make_pair(ns, nt, motion="fixed")
            out["literal_T_gt"] = c4_variant(R, R.local_rank, lsrc, ltgt, 0.15, lT, args.steps,
                                             "SURVEY 8d's T_gt = R_y(5 deg) R_x(1 deg), t = (0.02, -0.01, 0.015) on "
                                             "S-surf %d -> %d, radius 0.15" % (ns, nt))
            out["value_literal_T_gt"] = out["literal_T_gt"]["from_initial_pose"]["icp_iterations_per_sec"]
            if R.world == 1 and not args.no_cpu_baseline:
                # the compiled reference on the same pair and radius (its KD-tree does not care about the radius), one sample
                lcb = cpu_baseline_c4(lsrc, ltgt, 0.15, args.cpu_iters, 1)
                lc = _lib.Context(R.local_rank)
                lc.set_clouds_f64(lsrc, ltgt)
                lTg = lc.run(None, 0.15, 1 + args.cpu_iters, 0.0, 0.0)
                lcb["gpu_vs_cpu_rel_frobenius"] = synth.rel_frobenius(lTg.transformation_, np.array(lcb.pop("T")))
                lcb["gpu_search_kernel"] = lc.search_kernel_used()
                lc.close()
                out["literal_T_gt"]["cpu_baseline"] = lcb
            del lsrc, ltgt
            lsrc, ltgt, lT, _ = synth.make_pair(16384, 65536, motion="fixed")
            out["literal_T_gt_small"] = c4_variant(R, R.local_rank, lsrc, ltgt, 0.15, lT, args.steps,
                                                   "the same motion on S-surf 16,384 -> 65,536, radius 0.15")
            del lsrc, ltgt
        if brute is not None:
            b = brute_roofline(ns_local, nt_local, brute["nn_ms"], tile, load_traffic("brute", ns_local, nt_local))
            b.update(steps=brute["steps"], ms_per_step=brute["ms_per_step"],
                     iterations_per_sec=1e3 / brute["ms_per_step"],
                     mpairs_per_sec=float(ns) * nt / brute["ms_per_step"] / 1e3,
                     rel_frobenius_vs_grid=brute["rel_frobenius_vs_grid"])
            out["brute_force"] = b
        if f32_extra is not None:
            out["f32_search"] = f32_extra
        if weak is not None:
            out["weak_scaling"] = weak
        if not args.no_extras:
            sw = {"c4_grid_strong": {"icp_iterations_per_sec": args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
                                     "expectation": "one iteration of ~37 us of which ~14 us are launch, block reduction, "
                                                    "fold and host turn-around that no shard shortens: halving the queries "
                                                    "per rank does not halve it (DESIGN.md 5: 1.19x / 1.37x / 1.50x at "
                                                    "2 / 4 / 8 GPUs measured at the per-rank sizes, before the exchange "
                                                    "costs anything)"}}
            sw["c4_weak"] = ({"icp_iterations_per_sec": weak["icp_iterations_per_sec"], "ms_per_step": weak["ms_per_step"],
                              "source_points": weak["ns"]} if weak is not None else
                             {"icp_iterations_per_sec": args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
                              "source_points": ns})
            if large is not None:
                sw["c4_large_strong"] = large
            if c5 is not None:
                sw["c5_replicas"] = c5
            out["scaling_workloads"] = sw
        if R.world == 1 and not args.no_extras and mode == "grid":
            out["end_to_end"] = {"c4": c4_end_to_end(R.local_rank, src, tgt, radius),
                                 "c2_5k_20k": c4_end_to_end(R.local_rank, *synth.make_pair(5000, 20000)[:2], 0.075, iters=20),
                                 "c2_5k_20k_yaw_sweep_24": yaw_sweep_end_to_end(R.local_rank, *synth.make_pair(5000, 20000)[:2], 0.075),
                                 "note": "host arrays in -> transformation out on a context whose buffers exist: both "
                                         "clouds cross PCIe as the caller's f64 values; never `value`"}
            out["roofline_saturated"] = c4_saturated(R.local_rank, tgt, nt, radius)
        if R.world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline_c4(src, tgt, radius, args.cpu_iters, args.cpu_repeats)
            Tg = ctx.run(None, radius, 1 + args.cpu_iters, 0.0, 0.0)       # parity on this workload, same iteration count
            cb["gpu_vs_cpu_rel_frobenius"] = synth.rel_frobenius(Tg.transformation_, np.array(cb.pop("T")))
            out["cpu_baseline"] = cb
    # N > 1, opt-in mode: the ranks' launches kept alive across passes (VISMA_ICP_PERSIST_RANKS=1: the folding workgroups
    # of the ranks' persistent launches meet in the mailboxes pass after pass) -- measured AFTER `value`, which uses the
    # default (one launch per pass), under a watchdog: should this mode hang on first contact with real peers, rank 0
    # still prints the line (with the error in it) and every rank leaves
    if (R.dist is not None and args.shard == "source" and "mailboxes" in comm_kind and not args.no_extras
            and os.environ.get("VISMA_ICP_PERSIST_RANKS") != "1" and mode == "grid"):
        res = ranks_persistent_probe(R, args, src, tgt, ns, nt, radius, out)
        if out is not None:
            out["ranks_persistent"] = res
    ctx.close()
    return out


def ranks_persistent_probe(R, args, src, tgt, ns, nt, radius, out):
    import threading
    limit = float(os.environ.get("VISMA_BENCH_RANKS_PERSIST_LIMIT_S", "60"))

    def fire():
        if R.rank == 0 and out is not None:
            out["ranks_persistent"] = {"error": "no result within %.0f s: the watchdog printed this line and ended the ranks" % limit}
            emit(out, args.extras_file)
        else:
            time.sleep(2.0)
        os._exit(0)
    wd = threading.Timer(limit, fire)
    wd.daemon = True
    wd.start()
    res = None
    os.environ["VISMA_ICP_PERSIST_RANKS"] = "1"
    try:
        pctx, _, _ = c4_context(R, args, src, tgt, ns, nt)
        pctx, kind = attach_comm(R, pctx, lambda: c4_context(R, args, src, tgt, ns, nt)[0], args)
        _, plast, pel, ptm, _ = timed_registrations(R, pctx, radius, args.warmup, args.steps, args.nn, 4, 3)
        pe = float(np.median(pel))
        res = {"icp_iterations_per_sec": args.steps / pe, "ms_per_step": pe / args.steps * 1e3, "median_of": len(pel),
               "transport": kind, "fitness": plast.fitness_,
               "launch_mode": ("ONE persistent launch per host loop on every rank" if ptm.get("persist_passes", 0) > 0
                               else "one launch per pass (the ranks share a device, or the launch did not fit)"),
               "persist_passes_timed": ptm.get("persist_passes", 0), "persist_aborts": ptm.get("persist_aborts", 0)}
        pctx.close()
    except Exception as e:      # noqa: BLE001
        res = {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        wd.cancel()
        os.environ.pop("VISMA_ICP_PERSIST_RANKS", None)
    return res


# ------------------------------------------------------------------------------------------
# workloads c3 / c5: independent problems, replicas only
# ------------------------------------------------------------------------------------------
def c3_problems():
    """12 objects x 24 yaw starts: source 4k..40k points, target half of it, r = 0.02."""
    from visma_amd import synth
    rng = np.random.default_rng(3)
    objs = []
    for i in range(12):
        ns = int(rng.integers(4000, 40000))
        src, tgt, _, _ = synth.make_pair(ns, ns // 2, seed_t=300 + i, seed_s=400 + i)
        objs.append((src, tgt))
    probs = []
    for oi, (src, tgt) in enumerate(objs):
        for k in range(24):
            probs.append((src, tgt, synth.make_T(synth.rot_y(2 * np.pi * k / 24), [0, 0, 0]), 0.02, oi))
    return objs, probs


def run_c3(R, args):
    from visma_amd import _lib, synth
    objs, probs = c3_problems()
    # static deal: objects (with their 24 yaw starts, which share clouds) sorted by size, round-robin
    order = sorted(range(len(objs)), key=lambda i: -len(objs[i][0]) * len(objs[i][1]))
    mine_order = order[R.rank::R.world]
    mine = set(mine_order)
    # workers per GPU: the rank's objects dealt (by size, round-robin) to W contexts, each with its own stream; their
    # batches run side by side -- while one packs, uploads or runs its one-thread solves, the others' searches have the GPU
    W = max(1, min(int(os.environ.get("VISMA_C3_WORKERS_PER_GPU", "2")), len(mine_order)))
    ctxs = [_lib.Context(R.local_rank) for _ in range(W)]
    ctx = ctxs[0]
    iters = 30
    my = [p[:4] for p in probs if p[4] in mine]
    batch = ctx.make_batch(my)                  # the C array of problems, built once (not part of a step)

    def one_step():
        # visma_icp_run_batch_multi: problems over the same target stay together, the groups go to the W worker
        # contexts by size, every worker runs its share as one batch on its own host thread and stream
        return _lib.run_batch_multi(ctxs, batch, max_iter=iters)

    for _ in range(max(args.warmup, 1)):
        one_step()
    R.barrier_sync()
    t0 = time.perf_counter()
    its = 0
    for _ in range(args.steps):
        res = one_step()
        its += sum(r.iterations for r in res)
    R.barrier_sync()
    elapsed = R.reduce_max(time.perf_counter() - t0)
    # the search launches of the WHOLE batch on one context, alone on the GPU, timed by HIP events (after the timed
    # region)
    ctx.set_profiling(1)
    ctx.get_timing(reset=True)
    ctx.run_batch(batch, max_iter=iters)
    tm = ctx.get_timing(reset=True)
    ctx.set_profiling(0)
    my_queries = sum(len(p[0]) for p in my)
    total_its = R.reduce_sum(float(its))
    nl = max(tm["nn_launches"], 1)
    nn_ms = R.reduce_max(tm["nn_ms"] / nl)
    out = None
    if R.rank == 0:
        queries = my_queries
        nt_total = sum(len(objs[i][1]) for i in mine)
        roofline = grid_roofline(queries, nt_total, nn_ms, tm["grid_candidates"] / nl, tm["grid_candidates_27cell"] / nl,
                                 batch_traffic("c3:wave") if R.world == 1 else None, True,
                                 "wave" if ctx.search_kernel_used() == "warm" else "serial", 0.0)   # (batches: grid_wave.hip)
        roofline["traffic_source"] = "profiles/traffic.json c3:wave (PMC passes of `bench.py --workload c3` with one worker context, all " \
                                     "problems per launch; not this run)"
        roofline["launches_timed"] = tm["nn_launches"]
        out = {
            "metric": "icp_iterations_per_sec", "value": total_its / elapsed, "unit": "ICP iterations/s",
            "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3 all objects of one scene in flight: 12 objects x 24 yaw starts = 288 ICPs, "
                                   "source 4k..40k -> target half, r=0.02, <= 30 iterations each",
                       "problems": len(probs), "source_points_per_pass": sum(len(p[0]) for p in probs),
                       "search": ctx.search_mode_used(),
                       "parallelism": "replicas only: objects dealt round-robin by size over %d rank(s) x %d worker context(s) "
                                      "per GPU (visma_icp_run_batch_multi: each its own stream and host thread, shares side by side), no collective" % (R.world, W)},
            "config_short": {"parallelism": "replicas only: %d rank(s) x %d worker context(s) per GPU, no collective" % (R.world, W)},
            "problems_per_sec": len(probs) * args.steps / elapsed,
            "roofline": roofline,
        }
        if R.world == 1 and not args.no_cpu_baseline:
            kind, threads, run = cpu_registration()
            t0 = time.perf_counter()
            n_it, worst = 0, 0.0
            sample = probs[::24][:6]                     # the yaw-0 start of six objects
            for src, tgt, init, r, _ in sample:
                w = run(src, tgt, r, iters, init)
                n_it += iters
            dt = time.perf_counter() - t0
            one = _lib.Context(R.local_rank)
            for src, tgt, init, r, _ in sample[:3]:
                one.set_clouds_f64(src, tgt)
                g = one.run(init, r, iters, 0.0, 0.0)
                w = run(src, tgt, r, iters, init)
                worst = max(worst, synth.rel_frobenius(g.transformation_, w.T))
            one.close()
            out["cpu_baseline"] = {"value": n_it / dt, "unit": "ICP iterations/s", "cores": int(threads), "kind": kind,
                                   "sample": "6 of the 288 problems, %d iterations each, one after another "
                                             "(KD-tree builds included)" % iters,
                                   "gpu_vs_cpu_rel_frobenius": worst}
    for c in ctxs:
        c.close()
    return out


def c5_corpus():
    """Synthetic corpus: 12 scenes x 16 CAD candidates; a work item = one orientation-constrained registration
    (24 yaw starts, src/annotation.cpp:29-64) of a candidate against a scene."""
    from visma_amd import synth
    rng = np.random.default_rng(5)
    scenes = [synth.surface_points(int(rng.integers(15000, 40000)), 900 + s).astype(np.float32).astype(np.float64)
              for s in range(12)]
    cads = []
    for c in range(16):
        n = int(rng.integers(3000, 12000))
        p = synth.surface_points(n, 950 + c)
        Ti = np.linalg.inv(synth.make_T(synth.rot_y(rng.uniform(-0.05, 0.05)), rng.standard_normal(3) * 0.01))
        cads.append((p @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32).astype(np.float64))
    items = [(s, c) for s in range(len(scenes)) for c in range(len(cads))]
    return scenes, cads, items


C5_CHUNK = int(os.environ.get("VISMA_C5_CHUNK", "16"))         # work items a worker takes from the counter at a time: 16 x 24 = 384 ICPs in flight per launch
                      # (measured on one MI355X, round 2, one worker: 1 item per launch 167 k iterations/s, 2: 326 k, 4: 482 k,
                      #  8: 630 k; round 3, three workers per GPU: 4: 1.80 M, 8: 2.02 M, 12: 2.08 M, 16: 2.17 M, 24: 2.02 M)


def c5_chunk_problems(scenes, cads, chunk_items, radius, level):
    """The 24 yaw starts (src/annotation.cpp:35-39) of every item of a chunk as one batch: problems that pass
    the same arrays share one upload and one grid inside visma_icp_run_batch."""
    from visma_amd import synth
    probs = []
    for s, c in chunk_items:
        for k in range(level):
            probs.append((cads[c], scenes[s], synth.make_T(synth.rot_y(2 * np.pi * k / level), [0, 0, 0]), radius))
    return probs


def c5_pick(results, level):
    """feh::RegisterModelToScene's choice per item: the first start with strictly the most correspondences."""
    out = []
    for a in range(0, len(results), level):
        per = results[a:a + level]
        best = max(range(level), key=lambda k: (per[k].num_correspondences, -k))
        out.append((best, per[best]))
    return out


def run_c5(R, args, tag=""):
    """The corpus through the library's NATIVE work queue (visma_icp_run_corpus): this process is one rank with
    one context; with several ranks (one process per GPU) the counter lives in shared memory and every rank's
    host thread pulls chunks from it with atomic adds."""
    import ctypes
    from multiprocessing import shared_memory
    from visma_amd import _lib
    scenes, cads, items = c5_corpus()
    radius, level, iters = 0.05, 24, 30
    ctx = _lib.Context(R.local_rank)
    # queue workers per GPU (visma_icp_run_corpus takes any number of contexts; each has its own stream): while one
    # worker packs and uploads its next chunk -- and while its one-thread solves run -- the other's searches have the GPU
    nctx = max(1, int(os.environ.get("VISMA_C5_WORKERS_PER_GPU", "4")))
    ctxs = [ctx] + [_lib.Context(R.local_rank) for _ in range(nctx - 1)]
    corpus = _lib.Corpus([(cads[c], scenes[s]) for s, c in items], level=level, max_dist=radius, max_iter=iters,
                         rel_fitness=1e-6, rel_rmse=1e-6, chunk=C5_CHUNK)
    nwarm = max(args.warmup, 1)
    nslots = nwarm + args.steps
    shm = None
    if R.dist is not None:
        # one counter per pass (a rank may start the next pass while another still works on this one)
        name = "visma_c5_%s_%s" % (os.environ.get("MASTER_PORT", "0"), tag or "w")
        if R.rank == 0:
            try:
                shared_memory.SharedMemory(name=name).unlink()            # left behind by a killed run
            except Exception:      # noqa: BLE001
                pass
            shm = shared_memory.SharedMemory(name=name, create=True, size=8 * nslots)
            shm.buf[:8 * nslots] = bytes(8 * nslots)
        R.dist.barrier()
        if R.rank != 0:
            shm = shared_memory.SharedMemory(name=name)
            try:
                # (rank 0 owns the segment and unlinks it: keep this process's resource tracker from doing -- and
                #  complaining about -- the same at exit)
                from multiprocessing import resource_tracker
                resource_tracker.unregister(shm._name, "shared_memory")
            except Exception:      # noqa: BLE001
                pass
        base = ctypes.addressof(ctypes.c_char.from_buffer(shm.buf))

    def one_pass(slot):
        res = corpus.run(ctxs, (base + 8 * slot) if shm is not None else None)
        mine = [r for r in res if r[2] >= 0]
        return sum(r[3] for r in mine), len(mine)

    for w in range(nwarm):
        one_pass(w)
    R.barrier_sync()
    t0 = time.perf_counter()
    its = done = 0
    for step in range(args.steps):
        a, b = one_pass(nwarm + step)
        its += a
        done += b
    R.barrier_sync()
    elapsed = R.reduce_max(time.perf_counter() - t0)
    total_its = R.reduce_sum(float(its))
    per_rank_items = done
    total_items = int(round(R.reduce_sum(float(done))))
    if shm is not None:
        del base
        R.dist.barrier()
        shm.close()
        if R.rank == 0:
            shm.unlink()
    # roofline of the batch kernel: one profiled pass over the first chunks (after the timed region)
    roofline = None
    if R.rank == 0:
        ctx.set_profiling(1)
        b_alg = b_comp = ms = 0.0
        launches = 0
        for a in range(0, 8, C5_CHUNK):
            chunk = items[a:a + C5_CHUNK]
            ctx.get_timing(reset=True)
            ctx.run_batch(c5_chunk_problems(scenes, cads, chunk, radius, level), max_iter=iters)
            tm = ctx.get_timing(reset=True)
            nl = tm["nn_launches"]
            q = sum(len(cads[c]) for s, c in chunk) * level
            b_alg += warm_bytes(nl * q, tm["grid_certified"], tm["grid_candidates_27cell"], tm["grid_candidates"])
            b_comp += nl * (sum(len(scenes[s]) for s, c in chunk) * 12.0 + q * 72.0)
            ms += tm["nn_ms"]
            launches += nl
        ctx.set_profiling(0)
        if ms > 0:
            gbps = b_alg / (ms * 1e-3) / 1e9
            roofline = {"kernel": "nn_wave_kernel (grid_wave.hip) after each batch's first pass (%d problems per launch: %d items x 24 starts, exact search, fold fused)" % (
                            C5_CHUNK * level, C5_CHUNK),
                        "bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                        "traffic": batch_traffic("c5:wave") if R.world == 1 else None,
                        "traffic_source": "profiles/traffic.json c5:wave (PMC passes of `bench.py --workload c5`, one worker context)",
                        "avg_launch_ms": ms / max(launches, 1), "alg_bytes_per_launch": b_alg / max(launches, 1),
                        "compulsory_bytes": b_comp / max(launches, 1), "launches_timed": launches,
                        "note": "small clouds (3k-12k sources against 15k-40k targets, 384 problems per launch): every "
                                "launch is a hundred microseconds and its clouds stay in the L2 / Infinity Cache -- `achieved` "
                                "and `frac` are the kernel-counted EXAMINED bytes over the launch time, i.e. cache-resident "
                                "bandwidth quoted against the HBM peak, NOT HBM traffic; `traffic` / `traffic_frac` are what the "
                                "fabric counters saw"}
            if roofline["traffic"]:
                roofline["traffic_frac"] = roofline["traffic"] / (roofline["avg_launch_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS
    out = None
    if R.rank == 0:
        out = {
            "metric": "icp_iterations_per_sec", "value": total_its / elapsed, "unit": "ICP iterations/s",
            "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C5 corpus: %d scenes x %d CAD candidates = %d orientation-constrained registrations "
                                   "(24 yaw starts each, <= %d iterations, r=%.3g) per pass" % (
                                       len(scenes), len(cads), len(items), iters, radius),
                       "items": len(items), "search": ctx.search_mode_used(),
                       "parallelism": "replicas only: %d rank(s) pull chunks of %d work items from one atomic counter%s "
                                      "(visma_icp_run_corpus: native work queue, %d worker context(s) per GPU), no collective" % (
                                          R.world, C5_CHUNK, " in shared memory" if R.world > 1 else "", nctx)},
            "registrations_per_sec": len(items) * args.steps / elapsed,
            "items_done_by_rank0": per_rank_items, "items_done_by_all_ranks": total_items,
            "roofline": roofline,
        }
        if R.world == 1 and not args.no_cpu_baseline:
            kind, threads, run = cpu_registration()
            from visma_amd import synth
            t0 = time.perf_counter()
            n_it = 0
            for s, c in items[:2]:
                for k in range(0, level, 6):                          # 4 of the 24 yaw starts of two items
                    run(cads[c], scenes[s], radius, iters, synth.make_T(synth.rot_y(2 * np.pi * k / level), [0, 0, 0]))
                    n_it += iters
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": n_it / dt, "unit": "ICP iterations/s", "cores": int(threads), "kind": kind,
                                   "sample": "8 of the %d yaw starts (4 each of two work items), %d iterations each, "
                                             "KD-tree builds included" % (len(items) * level, iters)}
    for c in ctxs:
        c.close()
    return out


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1:
        respawn_under_torchrun(args.gpus)
    R = Ranks()
    if R.world != args.gpus:
        args.gpus = R.world
    from visma_amd import build
    if R.rank == 0:
        build.build_lib()
    if R.dist is not None:
        R.dist.barrier()
    out = {"c4": run_c4, "c3": run_c3, "c5": run_c5}[args.workload](R, args)
    if R.rank == 0 and out is not None:
        emit(out, args.extras_file)
    R.close()


if __name__ == "__main__":
    main()
