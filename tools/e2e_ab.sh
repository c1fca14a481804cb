mkdir -p gpurun_out/r6g
for i in 1 2 3; do
  for v in 0 1; do echo "== overlap=$v"; E2E_TRACE=0 VISMA_ICP_UPLOAD_OVERLAP=$v timeout 200 python tools/e2e_probe.py 2>&1 | grep "^ns="; done
done | tee gpurun_out/r6g/e2e_ab.txt
