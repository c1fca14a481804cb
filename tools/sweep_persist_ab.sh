mkdir -p gpurun_out/r7a
timeout 900 python -m pytest tests/test_sweep_persistent.py -m gpu -x -q > gpurun_out/r7a/tests.log 2>&1; tail -15 gpurun_out/r7a/tests.log
for i in 1 2 3; do VISMA_ICP_SWEEP_PERSIST=0 timeout 120 python tools/sweep_ab.py; timeout 120 python tools/sweep_ab.py; done | tee gpurun_out/r7a/sweep_ab.txt
