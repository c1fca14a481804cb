#!/bin/bash
# usage: tools/pmc_traffic.sh <outdir> <env assignments...> -- fabric traffic of the search kernel of tools/run_c4_iterations.py
# (FETCH_SIZE, WRITE_SIZE and the L2 hit/miss counters in separate passes, as the pool requires)
out=$1; shift
mkdir -p "$out"; export TMPDIR=/tmp
groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum")
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  ( cd /tmp && env "$@" timeout 240 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out" -o "t$i" -- python /root/repo/tools/run_c4_iterations.py > "$out/t$i.log" 2>&1 ) || echo "group '$g' failed"
done
python /root/repo/tools/pmc_summarize.py "$out" nn_ | cut -c1-60,150-400
