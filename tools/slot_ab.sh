mkdir -p gpurun_out/r6x
timeout 900 python -m pytest tests/test_persistent.py tests/test_certificate.py tests/test_c4_exact.py tests/test_warm_coop.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/r6x/tests.log 2>&1; tail -2 gpurun_out/r6x/tests.log
timeout 600 python tools/fuzz_persist_vs_per_pass.py 40 11 > gpurun_out/r6x/fuzz.log 2>&1; tail -1 gpurun_out/r6x/fuzz.log
bash tools/ab_libs.sh gpurun_out/r6x/ab.jsonl 3 /root/repo/_ab_old/libvisma_icp_noslot.so product
