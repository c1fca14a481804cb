#!/bin/bash
# A/B of the batch path on the GPU box: config 3 with the per-pass solve in a launch of its own (VISMA_ICP_SOLVE_IN_FOLD=0,
# rounds 1-4) and in the fold epilogue of the search launch (1, round 5), one worker context so that the kernel trace
# is not two batches overlapping: bench value + rocprofv3 kernel stats of each.  usage: tools/c3_solve_ab.sh <outdir>
out=${1:-/root/repo/gpurun_out/r05}
mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
for v in 0 1; do
  VISMA_C3_WORKERS_PER_GPU=1 VISMA_ICP_SOLVE_IN_FOLD=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c3prof$v" -o c3 -- python /root/repo/bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > "$out/c3prof$v.json" 2> "$out/c3prof$v.err" < /dev/null
  f=$(find "$out/c3prof$v" -name "*kernel_stats.csv" 2>/dev/null | head -1)
  echo "== SOLVE_IN_FOLD=$v"
  python -c "
import json;d=json.load(open('$out/c3prof$v.json'));print(d['value'],d['ms_per_step'])" < /dev/null
  if [ -n "$f" ]; then head -6 "$f" | cut -c1-60,300-420; cp "$f" "$out/c3_solve_in_fold_${v}_kernel_stats.csv"; fi
  rm -rf "$out/c3prof$v"
done
