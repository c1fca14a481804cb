#!/bin/bash
# usage: tools/pmc_traffic.sh <outdir> [ENV=.. ...]     fabric traffic of the search kernels of tools/run_c4_iterations.py,
#        PMC_CMD="python /root/repo/bench.py --workload c3 ..." tools/pmc_traffic.sh <outdir> [ENV=..]    ... of any command
# (FETCH_SIZE, WRITE_SIZE and the L2 hit/miss counters in separate passes, --kernel-trace only, as the pool requires;
#  the gfx950 corrections of MI355X_MICROARCH.md are applied by tools/collect_profiles.py)
out=$1; shift
mkdir -p "$out"; export TMPDIR=/tmp
cmd=${PMC_CMD:-python /root/repo/tools/run_c4_iterations.py}
groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum")
if [ -n "$PMC_FEW" ]; then groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"); fi
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  ( cd /tmp && env "$@" timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out" -o "t$i" -- $cmd > "$out/t$i.log" 2>&1 ) || echo "group '$g' failed"
  find "$out" -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
done
python /root/repo/tools/pmc_summarize.py "$out" nn_ | cut -c1-60,150-400
