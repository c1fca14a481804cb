#!/bin/bash
# Round profile set (run on the GPU box through gpurun; outputs under gpurun_out/prof_$1):
#   bench lines of the three workloads, rocprofv3 kernel stats of the default bench command,
#   PMC passes (separate runs, --kernel-trace only) incl. FETCH_SIZE / WRITE_SIZE for roofline.traffic
#   -- since round 5 for every regime the bench reports a roofline for: C4 from the identity (`value`), C4 converged,
#   partial overlap from the identity, 4 M queries, config 3, config 5
tag=${1:-r06}
out=/root/repo/gpurun_out/prof_$tag
mkdir -p $out; export TMPDIR=/tmp
cd /root/repo
timeout 900 python bench.py --steps 20 --warmup 5 --extras-file $out/bench_c4_extras.json > $out/bench_c4.json 2> $out/bench_c4.err < /dev/null
timeout 600 python bench.py --workload c3 --steps 5 --warmup 1 --extras-file $out/bench_c3_extras.json > $out/bench_c3.json 2> $out/bench_c3.err < /dev/null
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --extras-file $out/bench_c5_extras.json > $out/bench_c5.json 2> $out/bench_c5.err < /dev/null
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c4 -o c4 -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --extras-file $out/bench_c4_under_rocprof_extras.json > $out/bench_c4_under_rocprof.json 2> $out/rocprof_c4.err < /dev/null )
( cd /tmp && VISMA_C3_WORKERS_PER_GPU=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c3 -o c3 -- python /root/repo/bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --extras-file /tmp/c3_rocprof_extras.json > $out/bench_c3_under_rocprof.json 2> $out/rocprof_c3.err < /dev/null )
# fabric traffic, one regime per directory (tools/run_c4_iterations.py says what each runs)
PMC_FEW=1 VISMA_REGIME=initial tools/pmc_traffic.sh $out/pmc_initial VISMA_REGIME=initial > $out/pmc_initial.log 2>&1; python tools/pmc_summarize.py $out/pmc_initial nn_ > $out/pmc_traffic_initial_summary.csv
PMC_FEW=1 tools/pmc_traffic.sh $out/pmc > $out/pmc.log 2>&1; python tools/pmc_summarize.py $out/pmc nn_ > $out/pmc_traffic_summary.csv
PMC_FEW=1 tools/pmc_traffic.sh $out/pmc_partial VISMA_REGIME=initial VISMA_PAIR=partial > $out/pmc_partial.log 2>&1; python tools/pmc_summarize.py $out/pmc_partial nn_ > $out/pmc_traffic_partial_initial_summary.csv
PMC_FEW=1 tools/pmc_traffic.sh $out/pmc_sat VISMA_NS=4194304 > $out/pmc_sat.log 2>&1; python tools/pmc_summarize.py $out/pmc_sat nn_ > $out/pmc_traffic_saturated_summary.csv
PMC_FEW=1 PMC_TIMEOUT=300 PMC_CMD="python /root/repo/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --extras-file /tmp/pmc_extras.json" tools/pmc_traffic.sh $out/pmc_c3 VISMA_C3_WORKERS_PER_GPU=1 > $out/pmc_c3.log 2>&1; python tools/pmc_summarize.py $out/pmc_c3 nn_ > $out/pmc_traffic_c3_summary.csv
PMC_FEW=1 PMC_TIMEOUT=300 PMC_CMD="python /root/repo/bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --extras-file /tmp/pmc_extras.json" tools/pmc_traffic.sh $out/pmc_c5 VISMA_C5_WORKERS_PER_GPU=1 > $out/pmc_c5.log 2>&1; python tools/pmc_summarize.py $out/pmc_c5 nn_ > $out/pmc_traffic_c5_summary.csv
# instruction mix / L1 / wait counters of the default kernel (every pass under its own timeout), in the `value` regime
tools/pmc_quick.sh $out/pmc_kernel VISMA_REGIME=initial > $out/pmc_kernel_summary.txt 2>&1
python tools/pmc_summarize.py $out/pmc_kernel nn_ > $out/pmc_kernel_summary.csv
# the persistent launch: per-pass / per-workgroup clocks, host-side gaps, the mailbox round trips
( VISMA_ICP_PERSIST_TRACE=1 VISMA_ICP_PERSIST_TIMELINE=/tmp/tl_$tag.bin timeout 300 python tools/persist_probe.py 4194304 262144 5000 > $out/persist_probe_traced.jsonl 2> $out/persist_host_gaps.txt < /dev/null; python tools/persist_timeline.py /tmp/tl_$tag.bin > $out/persist_timeline.txt 2>&1 )
timeout 300 python tools/persist_probe.py 4194304 262144 131072 65536 5000 > $out/persist_probe.jsonl 2>&1 < /dev/null
timeout 300 python tools/cert_probe.py > $out/cert_probe.txt 2>&1 < /dev/null
mkdir -p tools/ubench/_build
for u in host_mailbox device_mailbox relay_latency; do
  [ -x tools/ubench/_build/$u ] || hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o tools/ubench/_build/$u 2> /dev/null
done
( timeout 60 tools/ubench/_build/host_mailbox; timeout 120 tools/ubench/_build/device_mailbox; timeout 60 tools/ubench/_build/relay_latency ) > $out/mailbox_ubench.txt 2>&1 < /dev/null
find $out -name "*kernel_trace.csv" -size +30M -delete
find $out -name "*.db" -delete
du -sh $out; ls $out
