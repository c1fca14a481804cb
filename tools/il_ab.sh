mkdir -p gpurun_out/r6v
timeout 900 python -m pytest tests/test_persistent.py tests/test_certificate.py tests/test_c4_exact.py tests/test_warm_coop.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/r6v/tests.log 2>&1; tail -2 gpurun_out/r6v/tests.log
bash tools/ab_libs.sh gpurun_out/r6v/ab.jsonl 3 /root/repo/_ab_old/libvisma_icp_noil.so product
