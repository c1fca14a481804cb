mkdir -p gpurun_out/r6k
timeout 1200 python -m pytest tests/test_persistent.py tests/test_certificate.py tests/test_c4_exact.py tests/test_gpu_golden.py tests/test_warm_coop.py -m gpu -x -q > gpurun_out/r6k/tests.log 2>&1; tail -3 gpurun_out/r6k/tests.log
timeout 600 python tools/fuzz_persist_vs_per_pass.py 40 7 > gpurun_out/r6k/fuzz.log 2>&1; tail -3 gpurun_out/r6k/fuzz.log
for i in 1 2 3; do for v in 0 1; do VISMA_ICP_PERSIST_EARLY=$v timeout 300 python tools/ab_probe.py 4194304 262144 65536 5000 2>&1 | grep '"ns"' | sed "s/^/early=$v /"; done; done | tee gpurun_out/r6k/ab.txt
