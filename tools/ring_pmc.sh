#!/bin/bash
# usage: tools/ring_pmc.sh <outdir> [env assignments...]  -- counters of the ring search's launches (tools/ring_iterations.py)
out=$1; shift
mkdir -p "$out"; export TMPDIR=/tmp
groups=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR"
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"
        "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr GRBM_GUI_ACTIVE"
        "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum")
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  ( cd /tmp && env "$@" timeout 240 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out" -o "g$i" -- python /root/repo/tools/ring_iterations.py > "$out/g$i.log" 2>&1 ) || echo "group '$g' failed"
done
python /root/repo/tools/pmc_summarize.py "$out" nn_ring
