mkdir -p gpurun_out/r6r
timeout 900 python -m pytest tests/test_persistent.py tests/test_certificate.py tests/test_c4_exact.py tests/test_warm_coop.py -m gpu -x -q > gpurun_out/r6r/tests.log 2>&1; tail -2 gpurun_out/r6r/tests.log
bash tools/ab_libs.sh gpurun_out/r6r/ab.jsonl 3 /root/repo/_ab_old/libvisma_icp_nopf.so product
