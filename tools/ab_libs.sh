# A/B of the C4 registration (tools/ab_probe.py: us per iteration 1..20 from the identity, 41..60) under several builds of
# the library, alternating on the same box.  usage: bash tools/ab_libs.sh <out-file> <rounds> product|<path.so> ...
out=$1; rounds=$2; shift 2
mkdir -p $(dirname $out)
for i in $(seq $rounds); do
  for lib in "$@"; do
    if [ $lib = product ]; then unset VISMA_ICP_LIB; else export VISMA_ICP_LIB=$lib; fi
    timeout 300 python tools/ab_probe.py 4194304 262144 2>&1 | tail -1 >> $out
  done
done
cat $out
