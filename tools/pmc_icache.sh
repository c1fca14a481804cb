out=/root/repo/gpurun_out/r3l; mkdir -p $out; export TMPDIR=/tmp
i=0
for g in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  ( cd /tmp && timeout 240 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o "i$i" -- python /root/repo/tools/run_c4_iterations.py > $out/i$i.log 2>&1 ) || echo "group $g failed"
done
