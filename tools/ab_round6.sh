# round 6 A/B on the GPU box: the product library against _ab_old/libvisma_icp_base.so (the library before a change),
# alternating runs on the same box.  usage: bash tools/ab_round6.sh <out-dir> <what: c3|c5|c4|tests ...>
out=${1:-gpurun_out/ab}; shift
mkdir -p $out
for what in "$@"; do
  case $what in
    tests) timeout 1500 python -m pytest tests/test_wave_certificate.py tests/test_c3_batch.py tests/test_c5_corpus.py tests/test_annotation.py tests/test_warm_coop.py tests/test_gpu_golden.py -m gpu -x -q > $out/tests.log 2>&1; tail -4 $out/tests.log ;;
    c3|c5) for i in 1 2; do
        for lib in base product; do
          if [ $lib = base ]; then export VISMA_ICP_LIB=/root/repo/_ab_old/libvisma_icp_base.so; else unset VISMA_ICP_LIB; fi
          steps=5; [ $what = c5 ] && steps=2
          timeout 600 python bench.py --workload $what --steps $steps --warmup 1 --no-cpu-baseline --extras-file /tmp/x.json 2>> $out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$what $lib', d['value'], d.get('registrations_per_sec'), d['roofline']['avg_launch_ms'], d['roofline'].get('certified_fraction'))" | tee -a $out/ab.txt
        done; done ;;
    c4) for i in 1 2 3; do
        VISMA_ICP_LIB=/root/repo/_ab_old/libvisma_icp_base.so timeout 300 python tools/ab_probe.py 4194304 262144 >> $out/ab_c4.jsonl 2>&1
        timeout 300 python tools/ab_probe.py 4194304 262144 >> $out/ab_c4.jsonl 2>&1
      done; cat $out/ab_c4.jsonl ;;
  esac
done
