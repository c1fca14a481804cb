mkdir -p gpurun_out/r6m
for i in 1 2 3; do
  timeout 300 python tools/ab_probe.py 4194304 262144 2>&1 | tail -1 | sed 's/^/base /'
  VISMA_ICP_LIB=/root/repo/_ab_old/libvisma_icp_ru.so VISMA_ICP_RUNNER_UP=1 timeout 300 python tools/ab_probe.py 4194304 262144 2>&1 | tail -1 | sed 's/^/ru /'
done | tee gpurun_out/r6m/ru_ab.txt
