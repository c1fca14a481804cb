# C3 / C5 bench values under several builds of the library, alternating on one box.
# usage: bash tools/batch_libs_ab.sh <out-file> <rounds> product|<lib.so> ...
out=$1; rounds=$2; shift 2
mkdir -p $(dirname $out)
for i in $(seq $rounds); do
  for lib in "$@"; do
    if [ $lib = product ]; then unset VISMA_ICP_LIB; else export VISMA_ICP_LIB=$lib; fi
    for w in c3 c5; do
      steps=5; [ $w = c5 ] && steps=2
      timeout 600 python bench.py --workload $w --steps $steps --warmup 1 --no-cpu-baseline --extras-file /tmp/x.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', '$(basename $lib)', round(d['value']), d.get('registrations_per_sec'), d['roofline']['avg_launch_ms'])" | tee -a $out
    done
  done
done
