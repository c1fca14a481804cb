#!/bin/bash
# usage: tools/_pmc.sh <outdir> <env assignments...>  -- counters of tools/run_c4_iterations.py
out=$1; shift
mkdir -p "$out"; export TMPDIR=/tmp
groups=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR"
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
        "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr GRBM_GUI_ACTIVE")
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  ( cd /tmp && env "$@" timeout 240 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out" -o "g$i" -- python /root/repo/tools/run_c4_iterations.py > "$out/g$i.log" 2>&1 ) || echo "group '$g' failed"
done
python /root/repo/tools/pmc_summarize.py "$out" nn_ | cut -c1-60,150-400
