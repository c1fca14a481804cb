#!/usr/bin/env python3
"""bench.py -- ICP iterations/s of the MI355X-native registration path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nn auto|grid|brute]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json metric "ICP iterations/sec + corresp. M-pairs/sec,
64k->4M-pt target"): synthetic S-surf clouds, source
262,144 points -> target 4,194,304 points, fp32 search / f64 statistics.
One STEP = one ICP iteration = one fused transform + nearest-neighbour pass of
every source point against the target, one Jacobian/residual reduction, one
host solve, T <- update*T.  Inputs are resident in HBM before the timed region
(the radius-cell grid is built once per target/radius, outside the timed
region, like the reference's KD-tree; its build time is reported).

NN search: `auto` (default) = the radius-cell uniform grid (exact radius-limited
1-NN; HBM-bound); `brute` = the LDS-tiled brute-force kernel north_star names
(fp32-VALU-bound).  Both give bit-identical correspondences.  A few brute-force
steps are always run after the timed region and reported under `brute_force`.

N > 1: the SOURCE is sharded across ranks (each rank holds the full target),
every rank reduces its shard to the 38 f64 normal-equation accumulators and
ONE ncclAllReduce (RCCL over xGMI) per iteration sums them; total work is
fixed, so "scaling" is "strong" and `value` is the job's iterations/s.

Prints ONE JSON line on rank 0 (contract in the task description), with the
extra objects `roofline` (dominant kernel = NN correspondence) and
`cpu_baseline` (the reference itself, oracle/_ref, timed on this host).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

NS_DEFAULT = 262144
NT_DEFAULT = 4194304
PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
PEAK_FP32_TFLOPS = 157.3        # ... FP32 vector == FP32 (f32-input) MFMA dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ns", type=int, default=NS_DEFAULT)
    ap.add_argument("--nt", type=int, default=NT_DEFAULT)
    ap.add_argument("--nn", choices=["auto", "grid", "brute"], default="auto")
    ap.add_argument("--shard", choices=["source", "target"], default="source",
                    help="multi-GPU decomposition: source points (one all-reduce of 38 f64 per iteration; "
                         "default) or target points (north_star's wording: MIN all-reduce of NS keys + the "
                         "same sum; for targets that exceed one GPU)")
    ap.add_argument("--brute-steps", type=int, default=3)
    ap.add_argument("--f64-steps", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=20)
    return ap.parse_args()


def cpu_baseline(src, tgt, radius, iters):
    """Time the CPU path on this host: the real reference (oracle/_ref, Open3D
    RegistrationICP with the KD-tree) when the prebuilt library is present,
    else our C restatement with its uniform-grid search ("port").
    Bounded sample: the SAME clouds, `iters` iterations; steady-state rate =
    (t(iters+1 its) - t(1 it)) / iters, so the one-off KD-tree build is reported
    separately and not charged to the iteration rate."""
    from oracle.oracle import Oracle, Ref
    threads = os.cpu_count() or 1
    if Ref.available():
        r = Ref()
        kind = "reference"

        def run(m):
            t0 = time.perf_counter()
            res = r.registration_icp(src, tgt, radius, max_iter=m, rel_fitness=0.0, rel_rmse=0.0)
            return time.perf_counter() - t0, res
    else:
        o = Oracle()
        kind = "port"
        threads = o.num_threads()

        def run(m):
            t0 = time.perf_counter()
            res = o.registration_icp(src, tgt, radius, max_iter=m, rel_fitness=0.0, rel_rmse=0.0, grid=True)
            return time.perf_counter() - t0, res
    t1, _ = run(1)
    t2, res = run(1 + iters)
    per_iter = max((t2 - t1) / iters, 1e-9)
    return {
        "value": 1.0 / per_iter, "unit": "ICP iterations/s", "cores": int(threads), "kind": kind,
        "sample": "same clouds %d->%d, r=%.4g: %d steady-state iterations "
                  "(t[%d its]-t[1 it]); setup (KD-tree build) + 2 passes %.2fs"
                  % (len(src), len(tgt), radius, iters, iters + 1, t1),
        "ms_per_iter": per_iter * 1e3, "setup_plus_first_iter_s": t1,
        "T": np.asarray(res.T).tolist(),
    }


def brute_roofline(ns_local, nt, nn_ms, tile, traffic):
    # ALGORITHMIC work of ONE brute-force NN launch on one rank:
    #   flops = 8 * NS_local * NT        (3 sub, 1 mul, 2 fma = 8 flop per pair)
    #   bytes = ceil(NS_local/S_TILE) * NT * 16  +  NS_local * 24
    flops = 8.0 * ns_local * nt
    s_tile = tile["block"] * (8 if ns_local >= 65536 else 2)
    b_alg = math.ceil(ns_local / s_tile) * nt * 16.0 + ns_local * 24.0
    tf = flops / (nn_ms * 1e-3) / 1e12
    return {
        "kernel": "nn_brute_kernel", "bound": "mfma", "achieved": tf, "peak": PEAK_FP32_TFLOPS,
        "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS, "traffic": traffic,
        "note": "fp32 compute roof: the brute-force pair loop is VALU-bound (~1e5 flop/B); gfx950's "
                "dense f32 MFMA peak equals its f32 vector peak (157.3 TF); the kernel issues VALU "
                "ops, no MFMA",
        "avg_launch_ms": nn_ms, "alg_flops_per_launch": flops,
        "pairs_per_launch": float(ns_local) * nt,
        "hbm_streamed": {"alg_bytes_per_launch": b_alg, "s_tile": s_tile,
                         "achieved_gbps": b_alg / (nn_ms * 1e-3) / 1e9,
                         "frac_of_8TBps": b_alg / (nn_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                         "compulsory_bytes": nt * 16.0 + ns_local * 24.0},
    }


def grid_roofline(ns_local, nt, nn_ms, cand_per_launch, cand27_per_launch, traffic):
    # ALGORITHMIC bytes of ONE grid launch on one rank: per query 16 B source +
    # 18 x 4 B cell-range lookups + 8 B (index, d2) out, plus 16 B per candidate
    # target point EXAMINED (counted by the kernel; rows of cells that provably
    # cannot hold a better candidate are skipped, so this is less than the full
    # 3x3x3 neighbourhood, whose byte count is given for reference).
    b_alg = ns_local * (16.0 + 72.0 + 8.0) + 16.0 * cand_per_launch
    b_27 = ns_local * (16.0 + 72.0 + 8.0) + 16.0 * cand27_per_launch
    gbps = b_alg / (nn_ms * 1e-3) / 1e9
    return {
        "kernel": "nn_grid_reduce_kernel", "bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS,
        "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS, "traffic": traffic,
        "avg_launch_ms": nn_ms, "alg_bytes_per_launch": b_alg,
        "candidates_per_query": cand_per_launch / max(ns_local, 1),
        "full_27cell": {"bytes_per_launch": b_27, "gbps": b_27 / (nn_ms * 1e-3) / 1e9,
                        "candidates_per_query": cand27_per_launch / max(ns_local, 1)},
        "compulsory_bytes": nt * 16.0 + ns_local * 24.0,
        "note": "fused transform + grid NN + Jacobian/residual reduction: ~10 dependent memory round trips "
                "per query (source, 18 cell bounds, row after row, winner); PMC: texture-address unit 45 % "
                "busy, L1 tag rate 53 % of its measured ceiling, HBM+MALL 40 %, VALU 30 % -- no unit saturated",
    }


def load_traffic(kind, ns_local, nt):
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tpath))
        return tj.get("%s:%dx%d" % (kind, ns_local, nt), {}).get("hbm_bytes_per_nn_launch")
    except Exception:
        return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d`"
                             % (args.gpus, args.gpus))
        args.gpus = world

    force_comm = os.environ.get("VISMA_ICP_FORCE_COMM") == "1"   # exercise RCCL even at N=1
    dist = None
    if world > 1 or force_comm:
        import torch
        import torch.distributed as dist
        # VISMA_BENCH_BACKEND=gloo: a dry run of the multi-rank logic on a box with fewer GPUs than ranks
        # (ranks share devices, torch tensors stay on the CPU, the exchange is the torch callback)
        backend = os.environ.get("VISMA_BENCH_BACKEND", "nccl")
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        tdev = "cuda" if backend == "nccl" else "cpu"

    from visma_amd import _lib, build, synth
    if rank == 0:
        build.build_lib()
    if dist is not None:
        dist.barrier()

    ns, nt = args.ns, args.nt
    src, tgt, T_gt, radius = synth.make_pair(ns, nt, motion="radius")

    ctx = _lib.Context(local_rank)
    # the library's default searches clouds of up to 131,072 source points in f64 (exact ties);
    # what counts here is the GLOBAL problem, not a rank's slice of it
    ctx.set_search_precision("auto" if ns <= 131072 else "f32")
    if args.shard == "source" or (world == 1 and not force_comm):
        # source shard of this rank (contiguous slice; full target everywhere)
        lo = (ns * rank) // world
        hi = (ns * (rank + 1)) // world
        ns_local, nt_local = hi - lo, nt
        # every rank must centre on the SAME point: set_clouds_f64 centres on the
        # (full) target centroid, which all ranks share.
        ctx.set_clouds_f64(src[lo:hi], tgt)
        ctx.set_global_source_count(ns)
    else:
        # target shard of this rank (contiguous slice of the global index space; all sources)
        lo = (nt * rank) // world
        hi = (nt * (rank + 1)) // world
        ns_local, nt_local = ns, hi - lo
        ctx.set_target_shard(lo, nt, tgt.mean(0))
        ctx.set_clouds_f64(src, tgt[lo:hi])
    comm_kind = "none"
    if dist is not None:
        import torch
        # the library's own RCCL communicator (dlopen'ed librccl): one ncclAllReduce of 38 f64 per
        # iteration on the context's stream.  Should it fail to come up on this node, every rank falls
        # back TOGETHER to the same exchange through torch.distributed (RCCL as well, but via a host
        # callback: slower) rather than leaving the job without a number.
        ok = 0 if (os.environ.get("VISMA_BENCH_FORCE_TORCH_COMM") == "1" or backend != "nccl") else 1
        uid_bytes = bytes(_lib.UNIQUE_ID_BYTES)
        if rank == 0 and ok:
            try:
                uid_bytes = _lib.comm_unique_id()
            except Exception as e:      # noqa: BLE001
                print("bench: ncclGetUniqueId failed (%s)" % e, file=sys.stderr)
                ok = 0
        uid = torch.tensor(list(uid_bytes), dtype=torch.uint8, device=tdev)
        dist.broadcast(uid, 0)
        flag = torch.tensor([ok], dtype=torch.int32, device=tdev)
        dist.broadcast(flag, 0)
        ok = int(flag.item())
        if ok:
            try:
                ctx.comm_init(rank, world, bytes(uid.cpu().tolist()))
            except Exception as e:      # noqa: BLE001
                print("bench: rank %d: ncclCommInitRank failed (%s)" % (rank, e), file=sys.stderr)
                ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()):
            comm_kind = "rccl"
        else:
            comm_kind = "torch.distributed callback"
            if args.shard == "target" and not (world == 1 and not force_comm):
                raise SystemExit("bench: the target-sharded mode needs the library's own RCCL communicator")
            ctx2 = _lib.Context(local_rank)                   # a context without the half-made communicator
            ctx2.set_search_precision("auto" if ns <= 131072 else "f32")
            ctx2.set_clouds_f64(src[(ns * rank) // world:(ns * (rank + 1)) // world], tgt)
            ctx2.set_global_source_count(ns)
            ctx = ctx2

            def torch_allreduce(a):
                t = torch.from_numpy(a.copy()).to(tdev)
                dist.all_reduce(t)
                a[:] = t.cpu().numpy()
            ctx.set_allreduce(torch_allreduce, rank, world)

    def sync_all():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_max(x):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ctx.set_nn_mode({"auto": _lib.NN_AUTO, "grid": _lib.NN_GRID, "brute": _lib.NN_BRUTE}[args.nn])
    # HIP-event timing of the kernels: every launch with brute force (117 ms each), every
    # 4th ICP pass with the grid (four event records cost ~14 us of a ~70 us iteration)
    prof_every = 1 if args.nn == "brute" else 4
    ctx.set_profiling(prof_every)
    T = np.eye(4)
    ctx.get_timing(reset=True)
    if args.warmup > 0:
        T, _ = ctx.iterate(T, radius, args.warmup)       # also builds the grid (one-off)
    setup = ctx.get_timing(reset=True)
    mode = "grid" if ctx.nn_mode_used() == _lib.NN_GRID else "brute"
    sync_all()
    t0 = time.perf_counter()
    T, last = ctx.iterate(T, radius, args.steps)       # every step ends with a stream sync
    sync_all()
    elapsed = reduce_max(time.perf_counter() - t0)
    tm = ctx.get_timing(reset=True)
    nn_ms = reduce_max(tm["nn_ms"] / max(tm["nn_launches"], 1))
    cand = tm["grid_candidates"] / max(tm["nn_launches"], 1)
    cand27 = tm["grid_candidates_27cell"] / max(tm["nn_launches"], 1)

    # a few brute-force steps (outside the timed region) for the north_star kernel's own numbers
    brute = None
    if mode != "brute" and args.brute_steps > 0:
        ctx.set_nn_mode(_lib.NN_BRUTE)
        ctx.set_profiling(1)
        ctx.iterate(T, radius, 1)
        ctx.get_timing(reset=True)
        tb0 = time.perf_counter()
        Tb, _ = ctx.iterate(np.eye(4), radius, args.brute_steps)
        tb = time.perf_counter() - tb0
        tmb = ctx.get_timing(reset=True)
        # same answer as the grid from the same start (bit-identical correspondences)
        Tg, _ = (ctx.set_nn_mode(_lib.NN_GRID), ctx.iterate(np.eye(4), radius, args.brute_steps))[1]
        brute = {"steps": args.brute_steps, "ms_per_step": tb / args.brute_steps * 1e3,
                 "nn_ms": reduce_max(tmb["nn_ms"] / max(tmb["nn_launches"], 1)),
                 "rel_frobenius_vs_grid": synth.rel_frobenius(Tb, Tg)}
        ctx.set_nn_mode({"auto": _lib.NN_AUTO, "grid": _lib.NN_GRID}[args.nn])

    # the same workload with the double-precision search (outside the timed region; 1 GPU only):
    # what exact tie-breaking would cost at this size
    f64_extra = None
    if world == 1 and mode == "grid" and not ctx.search_is_f64() and args.f64_steps > 0:
        c64 = _lib.Context(local_rank)
        c64.set_search_precision("f64")
        c64.set_clouds_f64(src, tgt)
        c64.set_nn_mode(_lib.NN_GRID)
        c64.set_profiling(1)
        T64w, _ = c64.iterate(np.eye(4), radius, 2)
        c64.get_timing(reset=True)
        t0f = time.perf_counter()
        T64, _ = c64.iterate(np.eye(4), radius, args.f64_steps)
        tf = time.perf_counter() - t0f
        tm64 = c64.get_timing(reset=True)
        Tf32, _ = ctx.iterate(np.eye(4), radius, args.f64_steps)
        f64_extra = {"steps": args.f64_steps, "iterations_per_sec": args.f64_steps / tf,
                     "nn_ms": tm64["nn_ms"] / max(tm64["nn_launches"], 1),
                     "rel_frobenius_vs_f32_search": synth.rel_frobenius(T64, Tf32)}
        del c64

    if rank == 0:
        tile = _lib.tile_config()
        if mode == "grid":
            roofline = grid_roofline(ns_local, nt_local, nn_ms, cand, cand27, load_traffic("grid", ns_local, nt_local))
        else:
            roofline = brute_roofline(ns_local, nt_local, nn_ms, tile, load_traffic("brute", ns_local, nt_local))
        roofline["launches_timed"] = tm["nn_launches"]
        roofline["timed_every_nth_pass"] = prof_every
        roofline["reduce_finalize_avg_ms"] = tm["reduce_ms"] / max(tm["reduce_launches"], 1)
        out = {
            "metric": "icp_iterations_per_sec", "value": args.steps / elapsed,
            "unit": "ICP iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if ctx.search_is_f64() else "f32", "data": "synthetic",
            "config": {"workload": "C4 S-surf %d-pt source -> %d-pt target, %d fixed ICP iterations, "
                                   "nn=%s" % (ns, nt, args.steps, mode),
                       "ns": ns, "nt": nt, "radius": radius, "solver": "kabsch", "nn": mode,
                       "parallelism": ("source-sharded x%d, 1 ncclAllReduce(38 f64)/iter%s" % (
                           world, "" if comm_kind in ("rccl", "none") else " [" + comm_kind + "]"))
                       if args.shard == "source" else
                       ("target-sharded x%d, ncclAllReduce(min, %d u64) + ncclAllReduce(38 f64)/iter" % (world, ns))},
            "mpairs_per_sec": float(ns) * nt * args.steps / elapsed / 1e6,
            "matched_corr_per_sec": last.num_correspondences * args.steps / elapsed,
            "fitness": last.fitness_, "inlier_rmse": last.inlier_rmse_,
            "err_vs_T_gt": synth.rel_frobenius(T, T_gt),
            "setup_ms": {"grid_build_kernels": setup["aux_ms"]},
            "roofline": roofline,
        }
        if brute is not None:
            b = brute_roofline(ns_local, nt_local, brute["nn_ms"], tile, load_traffic("brute", ns_local, nt_local))
            b.update(steps=brute["steps"], ms_per_step=brute["ms_per_step"],
                     iterations_per_sec=1e3 / brute["ms_per_step"],
                     mpairs_per_sec=float(ns) * nt / brute["ms_per_step"] / 1e3,
                     rel_frobenius_vs_grid=brute["rel_frobenius_vs_grid"])
            out["brute_force"] = b
        if f64_extra is not None:
            out["f64_search"] = f64_extra
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(src, tgt, radius, args.cpu_iters)
            # parity of the two paths on this workload, same iteration count
            Tg = ctx.run(None, radius, 1 + args.cpu_iters, 0.0, 0.0).transformation_
            cb["gpu_vs_cpu_rel_frobenius"] = synth.rel_frobenius(Tg, np.array(cb.pop("T")))
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        ctx.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
