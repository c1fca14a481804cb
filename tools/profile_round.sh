#!/bin/bash
# Round profile set (run on the GPU box through gpurun; outputs under gpurun_out/prof_$1):
#   bench lines of the three workloads, rocprofv3 kernel stats of the default bench command,
#   PMC passes (separate runs, --kernel-trace only) incl. FETCH_SIZE / WRITE_SIZE for roofline.traffic
tag=${1:-r04}
out=/root/repo/gpurun_out/prof_$tag
mkdir -p $out; export TMPDIR=/tmp
cd /root/repo
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_c4.json 2> $out/bench_c4.err
timeout 600 python bench.py --workload c3 --steps 5 --warmup 1 > $out/bench_c3.json 2> $out/bench_c3.err
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 > $out/bench_c5.json 2> $out/bench_c5.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c4 -o c4 -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_c4_under_rocprof.json 2> $out/rocprof_c4.err )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c3 -o c3 -- python /root/repo/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c3_under_rocprof.json 2> $out/rocprof_c3.err )
for g in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  t=$(echo "$g" | tr ' ' '_' | cut -c1-16)
  ( cd /tmp && timeout 240 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out/pmc -o $t -- python /root/repo/tools/run_c4_iterations.py > $out/pmc_$t.log 2>&1 )
done
python tools/pmc_summarize.py $out/pmc nn_ > $out/pmc_traffic_summary.csv
# the same counters with 4 M queries per launch (bench.py: roofline_saturated)
for g in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  t=$(echo "$g" | tr ' ' '_' | cut -c1-16)
  ( cd /tmp && VISMA_NS=4194304 timeout 240 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out/pmc_sat -o $t -- python /root/repo/tools/run_c4_iterations.py > $out/pmc_sat_$t.log 2>&1 )
done
python tools/pmc_summarize.py $out/pmc_sat nn_ > $out/pmc_traffic_saturated_summary.csv
# instruction mix / L1 / wait counters of the default kernel (every pass under its own timeout)
tools/pmc_quick.sh $out/pmc_kernel > $out/pmc_kernel_summary.txt 2>&1
python tools/pmc_summarize.py $out/pmc_kernel nn_ > $out/pmc_kernel_summary.csv
# the persistent launch: per-pass / per-workgroup clocks, host-side gaps, the two mailbox round trips
( VISMA_ICP_PERSIST_TRACE=1 VISMA_ICP_PERSIST_TIMELINE=/tmp/tl_$tag.bin timeout 300 python tools/persist_probe.py 4194304 262144 5000 > $out/persist_probe_traced.jsonl 2> $out/persist_host_gaps.txt; python tools/persist_timeline.py /tmp/tl_$tag.bin > $out/persist_timeline.txt 2>&1 )
timeout 300 python tools/persist_probe.py 4194304 262144 131072 65536 5000 > $out/persist_probe.jsonl 2>&1
mkdir -p tools/ubench/_build
for u in host_mailbox device_mailbox; do
  [ -x tools/ubench/_build/$u ] || hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o tools/ubench/_build/$u 2> /dev/null
done
( timeout 60 tools/ubench/_build/host_mailbox; timeout 120 tools/ubench/_build/device_mailbox ) > $out/mailbox_ubench.txt 2>&1
ls $out
