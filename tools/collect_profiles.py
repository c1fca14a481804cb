#!/usr/bin/env python3
"""Copy the judged summaries of tools/profile_round.sh from gpurun_out/prof_<tag>/ (scratch) into profiles/
(tracked) and refresh profiles/traffic.json from the PMC passes.   usage: collect_profiles.py [tag=r02]"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
pairs = [("bench_c4_extras.json", "%s_bench_c4_extras.json"), ("bench_c3_extras.json", "%s_bench_c3_extras.json"), ("bench_c5_extras.json", "%s_bench_c5_extras.json"),
         ("bench_c4_under_rocprof_extras.json", "%s_bench_c4_under_rocprof_extras.json"),
         ("bench_c4.json", "%s_bench_c4.json"), ("bench_c3.json", "%s_bench_c3.json"), ("bench_c5.json", "%s_bench_c5.json"),
         ("bench_c4_under_rocprof.json", "%s_bench_c4_under_rocprof.json"),
         ("stats_c4/c4_kernel_stats.csv", "%s_bench_c4_kernel_stats.csv"), ("stats_c3/c3_kernel_stats.csv", "%s_bench_c3_kernel_stats.csv"),
         ("pmc_traffic_summary.csv", "%s_c4_traffic_pmc_summary.csv"),
         ("pmc_traffic_initial_summary.csv", "%s_c4_from_identity_traffic_pmc_summary.csv"),
         ("pmc_traffic_partial_initial_summary.csv", "%s_c4_partial_overlap_from_identity_traffic_pmc_summary.csv"),
         ("pmc_traffic_c3_summary.csv", "%s_c3_traffic_pmc_summary.csv"), ("pmc_traffic_c5_summary.csv", "%s_c5_traffic_pmc_summary.csv"),
         ("cert_probe.txt", "%s_cert_probe.txt"),
         ("pmc_traffic_saturated_summary.csv", "%s_4m_queries_traffic_pmc_summary.csv"),
         ("pmc_kernel_summary.csv", "%s_c4_kernel_pmc_summary.csv"),
         ("persist_timeline.txt", "%s_persist_timeline.txt"), ("persist_host_gaps.txt", "%s_persist_host_gaps.txt"),
         ("persist_probe.jsonl", "%s_persist_probe.jsonl"), ("mailbox_ubench.txt", "%s_mailbox_ubench.txt")]
for a, b in pairs:
    p = os.path.join(src, a)
    if os.path.exists(p):
        shutil.copyfile(p, os.path.join(dst, b % tag))
        print("copied", b % tag)
    else:
        print("missing", p)
# The default bench command launches the search kernels at several sizes (C4, the saturated sizes, the C5 batches,
# the 5k -> 20k end-to-end case): rocprofv3's --stats averages them together, so the per-kernel durations are also
# tabulated by launch size from the kernel trace (262,144 threads = the C4 launches the roofline object times).
import collections
import statistics
kt = os.path.join(src, "stats_c4", "c4_kernel_trace.csv")
if os.path.exists(kt):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        n = r["Kernel_Name"]
        if "nn_coop" in n or "nn_grid_reduce" in n or "nn_brute" in n:
            acc[(n.split("(")[0][:60], int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = [("kernel", "grid_threads", "launches", "avg_ns", "median_ns", "min_ns", "max_ns")]
    for (n, g), v in sorted(acc.items(), key=lambda kv: -len(kv[1])):
        rows.append((n, g, len(v), round(sum(v) / len(v), 1), statistics.median(v), min(v), max(v)))
    csv.writer(open(os.path.join(dst, "%s_bench_c4_search_kernels_by_launch_size.csv" % tag), "w")).writerows(rows)
    print("wrote %s_bench_c4_search_kernels_by_launch_size.csv" % tag)

# Persistent launches last as long as their host loop (5-pass warm-up, the seven timed 20-pass blocks, 19-pass loops from
# the identity, ...): every launch of the C4 size in dispatch order, so that the seven the roofline object times -- the
# 2nd to 8th at 262,144 threads: they follow the warm-up's -- can be compared with its avg_launch_ms.
if os.path.exists(kt):
    rows = [("order_at_this_size", "grid_threads", "start_ns_since_first", "duration_ns")]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        if "nn_coop_kernel_persist" in r["Kernel_Name"]:
            per[int(r["Grid_Size_X"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for g in sorted(per, reverse=True):
        v = sorted(per[g])
        for i, (t, d) in enumerate(v):
            rows.append((i + 1, g, t - v[0][0], d))
    csv.writer(open(os.path.join(dst, "%s_bench_c4_persistent_launches_in_order.csv" % tag), "w")).writerows(rows)
    c4 = [d for _, d in sorted(per.get(262144, []))]
    if len(c4) >= 8:
        print("wrote %s_bench_c4_persistent_launches_in_order.csv; launches 2..8 at 262,144 threads: average %.1f us" % (tag, sum(c4[1:8]) / 7e3))

# traffic of the C4 kernels: FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md, HBM) + WRITE_SIZE, KiB -> bytes.
# tools/run_c4_iterations.py runs one cold pass (lane-serial kernel) and five warm-started ones.
p = os.path.join(src, "pmc_traffic_summary.csv")
tj_path = os.path.join(dst, "traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
for key, match, label in (("grid:262144x4194304", "nn_grid_reduce_kernel", "nn_grid_reduce_kernel<false,1,8,true,false,true> (lane-serial exact search: first pass of a registration; fold fused)"),
                          ("grid_warm:262144x4194304", "nn_coop_kernel_one", "nn_coop_kernel_one<false> (warm-started exact search with certificates, passes 28..47 of a registration; fold fused)")):
    vals = {}
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            # (the warm kernel: its last 20 dispatches -- the converged passes bench.py's `value` times)
            if match in r["kernel"] and (("[last" in r["kernel"]) == (match == "nn_coop_kernel_one")):
                vals[r["counter"]] = float(r["mean_per_dispatch"])
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        tj[key] = {
            "hbm_bytes_per_nn_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
            "fetch_size_kib_raw": vals["FETCH_SIZE"], "write_size_kib_raw": vals["WRITE_SIZE"],
            "correction": "FETCH_SIZE x2 (gfx950 wide-stream under-count, MI355X_MICROARCH.md HBM section); WRITE_SIZE "
                          "uncorrected; cross-check TCC_MISS_sum x 128 B = %.1f MB" % (vals.get("TCC_MISS_sum", 0.0) * 128 / 1e6),
            "kernel": label,
            "source": "profiles/%s_c4_traffic_pmc_summary.csv (rocprofv3 --pmc, one pass per counter group, tools/profile_round.sh)" % tag}
        print("traffic.json %s: %.1f MB per launch" % (key, tj[key]["hbm_bytes_per_nn_launch"] / 1e6))
# the persistent launch of the last 20 passes (one dispatch): per launch and per pass
vals = {}
if os.path.exists(p):
    for r in csv.DictReader(open(p)):
        if "nn_coop_kernel_persist" in r["kernel"] and "[last dispatch" in r["kernel"]:
            vals[r["counter"]] = float(r["mean_per_dispatch"])
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    b = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    tj["grid_persist:262144x4194304"] = {
        "hbm_bytes_per_nn_launch": b, "passes_per_launch": 20, "hbm_bytes_per_pass": b / 20.0,
        "fetch_size_kib_raw": vals["FETCH_SIZE"], "write_size_kib_raw": vals["WRITE_SIZE"],
        "write_bytes_per_pass": vals["WRITE_SIZE"] * 1024.0 / 20.0,
        "correction": "FETCH_SIZE x2 (gfx950 wide-stream under-count, MI355X_MICROARCH.md HBM section); WRITE_SIZE "
                      "uncorrected; cross-check TCC_MISS_sum x 128 B = %.1f MB; the polls of the command block (every "
                      "workgroup's first wave, system-scope loads of fine-grained device memory while the host solves) are in it"
                      % (vals.get("TCC_MISS_sum", 0.0) * 128 / 1e6),
        "kernel": "nn_coop_kernel_persist<false> (ONE launch running passes 28..47 of a registration: certificates, fold fused, "
                  "next transform through the command block)",
        "source": "profiles/%s_c4_traffic_pmc_summary.csv (rocprofv3 --pmc, one pass per counter group, tools/profile_round.sh)" % tag}
    print("traffic.json grid_persist: %.1f MB per launch of 20 passes = %.1f MB per pass" % (b / 1e6, b / 20e6))
# ... and with 4 M queries per launch
p = os.path.join(src, "pmc_traffic_saturated_summary.csv")
vals = {}
if os.path.exists(p):
    for r in csv.DictReader(open(p)):
        if "nn_coop_kernel" in r["kernel"] and "[last" in r["kernel"]:
            vals[r["counter"]] = float(r["mean_per_dispatch"])
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    tj["grid_warm:4194304x4194304"] = {
        "hbm_bytes_per_nn_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
        "fetch_size_kib_raw": vals["FETCH_SIZE"], "write_size_kib_raw": vals["WRITE_SIZE"],
        "correction": "as above; cross-check TCC_MISS_sum x 128 B = %.1f MB" % (vals.get("TCC_MISS_sum", 0.0) * 128 / 1e6),
        "kernel": "nn_coop_kernel_one<false>, 4,194,304 queries per launch",
        "source": "profiles/%s_4m_queries_traffic_pmc_summary.csv" % tag}
    print("traffic.json grid_warm 4M: %.1f MB per launch" % (tj["grid_warm:4194304x4194304"]["hbm_bytes_per_nn_launch"] / 1e6))


def summary_vals(path, match, suffix):
    """{counter: mean per dispatch} of the rows of a pmc_summarize.py table whose kernel contains `match` and whose label
    ends the way `suffix` says (None: the plain per-kernel rows)."""
    vals = {}
    if os.path.exists(path):
        for r in csv.DictReader(l for l in open(path) if "," in l):
            k = r.get("kernel", "")
            if match in k and ((suffix is None and "[" not in k) or (suffix is not None and suffix in k)):
                try:
                    vals[r["counter"]] = float(r["mean_per_dispatch"])
                except (KeyError, ValueError):
                    pass
    return vals


# round 5: the regime `value` is timed in -- a fresh registration of 20 iterations from the identity: the lane-serial cold
# pass (its own launch: traffic.json "grid:...") and ONE dispatch of the persistent kernel running the 19 warm passes --
# full overlap and partial overlap; and the batch kernels of config 3 / 5
PASSES = 19
for key, fn, what in (("grid_persist_initial:262144x4194304", "pmc_traffic_initial_summary.csv",
                       "ONE launch running the %d warm passes of a FRESH registration from the identity (bench.py `value`)" % PASSES),
                      ("grid_persist_partial_initial:262144x4194304", "pmc_traffic_partial_initial_summary.csv",
                       "the same on the partial-overlap pair (bench.py `partial_overlap.from_initial_pose`)")):
    vals = summary_vals(os.path.join(src, fn), "nn_coop_kernel_persist", "[last dispatch")
    cold = summary_vals(os.path.join(src, fn), "nn_grid_reduce_kernel", None)
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        b = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
        tj[key] = {"hbm_bytes_per_nn_launch": b, "passes_per_launch": PASSES, "hbm_bytes_per_pass": b / PASSES,
                   "fetch_size_kib_raw": vals["FETCH_SIZE"], "write_size_kib_raw": vals["WRITE_SIZE"],
                   "write_bytes_per_pass": vals["WRITE_SIZE"] * 1024.0 / PASSES,
                   "correction": "FETCH_SIZE x2 (gfx950 wide-stream under-count, MI355X_MICROARCH.md HBM section); WRITE_SIZE "
                                 "uncorrected; cross-check TCC_MISS_sum x 128 B = %.1f MB" % (vals.get("TCC_MISS_sum", 0.0) * 128 / 1e6),
                   "kernel": "nn_coop_kernel_persist<false>: " + what,
                   "source": "profiles/%s (rocprofv3 --pmc, one pass per counter group, tools/profile_round.sh)" % (
                       {"pmc_traffic_initial_summary.csv": "%s_c4_from_identity_traffic_pmc_summary.csv",
                        "pmc_traffic_partial_initial_summary.csv": "%s_c4_partial_overlap_from_identity_traffic_pmc_summary.csv"}[fn] % tag)}
        if "FETCH_SIZE" in cold and "WRITE_SIZE" in cold:
            tj[key]["cold_pass_launch_bytes"] = (2.0 * cold["FETCH_SIZE"] + cold["WRITE_SIZE"]) * 1024.0
        print("traffic.json %s: %.1f MB per pass (write %.2f MB per pass), cold pass launch %.1f MB" % (
            key, b / PASSES / 1e6, tj[key]["write_bytes_per_pass"] / 1e6, tj[key].get("cold_pass_launch_bytes", 0.0) / 1e6))
for key, fn, name in (("c3:wave", "pmc_traffic_c3_summary.csv", "%s_c3_traffic_pmc_summary.csv"),
                      ("c5:wave", "pmc_traffic_c5_summary.csv", "%s_c5_traffic_pmc_summary.csv")):
    vals = summary_vals(os.path.join(src, fn), "nn_wave_kernel_one", None)
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        tj[key] = {"hbm_bytes_per_nn_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
                   "fetch_size_kib_raw": vals["FETCH_SIZE"], "write_size_kib_raw": vals["WRITE_SIZE"],
                   "correction": "FETCH_SIZE x2, WRITE_SIZE uncorrected; cross-check TCC_MISS_sum x 128 B = %.1f MB" % (
                       vals.get("TCC_MISS_sum", 0.0) * 128 / 1e6),
                   "kernel": "nn_wave_kernel_one<false>: mean over every launch of one bench step (all problems of the batch per launch)",
                   "source": "profiles/" + name % tag}
        print("traffic.json %s: %.1f MB per launch" % (key, tj[key]["hbm_bytes_per_nn_launch"] / 1e6))
json.dump(tj, open(tj_path, "w"), indent=1)
