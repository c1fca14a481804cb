#!/bin/bash
# The large-radius regime's evidence (grid_ring.hip), one box: probes, policy table, end to end, rocprofv3 kernel stats and PMC
# counters of the ring kernel.  usage: bash tools/ring_profile.sh <absolute out dir>
out=$1; mkdir -p $out; export TMPDIR=/tmp
cd /root/repo
{
  echo "== tools/ring_probe.py: literal motion of SURVEY 8d, r = 0.15, 20 iterations from the identity; ring search (mode 1) / radius cells (mode 0)"
  timeout 200 python tools/ring_probe.py 262144 4194304 0.15 20 1
  timeout 300 python tools/ring_probe.py 65536 1048576 0.15 20 1,0
  timeout 200 python tools/ring_probe.py 16384 65536 0.15 20 1,0
  echo "== tools/ring_e2e.py: clouds in, transform out (context warm)"
  timeout 200 python tools/ring_e2e.py 262144 4194304 0.15 30
  timeout 200 python tools/ring_e2e.py 16384 65536 0.15 30
  echo "== tools/ring_policy_probe.py: registrations (30 iterations) and 24-start yaw sweeps, radius cells (mode 0) vs rings (mode 1)"
  timeout 600 python tools/ring_policy_probe.py
  echo "== lanes per query (VISMA_ICP_RING_LANES)"
  for L in 8 4 2 1; do echo "lanes $L"; VISMA_ICP_RING_LANES=$L timeout 200 python tools/ring_probe.py 262144 4194304 0.15 20 1 | cut -c150-480; done
} > $out/ring_search_probe.txt 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o ring -- python /root/repo/tools/ring_iterations.py > $out/stats.log 2>&1 )
bash tools/ring_pmc.sh $out/pmc > $out/ring_kernel_pmc_summary.csv 2>$out/pmc.err
ls $out $out/stats | head -30
