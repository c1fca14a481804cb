export TMPDIR=/tmp; cd /tmp
for v in 0 1; do
  VISMA_C3_WORKERS_PER_GPU=1 VISMA_ICP_SOLVE_IN_FOLD=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r05/c3prof$v -o c3 -- python /root/repo/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/r05/c3prof$v.json 2>/root/repo/gpurun_out/r05/c3prof$v.err
  f=$(find /root/repo/gpurun_out/r05/c3prof$v -name "*kernel_stats.csv" | head -1)
  echo "== SOLVE_IN_FOLD=$v"; python -c "
import json;d=json.load(open('/root/repo/gpurun_out/r05/c3prof$v.json'));print(d['value'],d['ms_per_step'])"
  head -6 $f | cut -c1-60,300-420
  find /root/repo/gpurun_out/r05/c3prof$v -name "*kernel_trace.csv" -delete
  find /root/repo/gpurun_out/r05/c3prof$v -name "*.db" -delete
done
