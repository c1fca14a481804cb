#!/bin/bash
# usage: tools/pmc_mem.sh <outdir> <env assignments...> -- texture-path counters of tools/run_c4_iterations.py
out=$1; shift
mkdir -p "$out"; export TMPDIR=/tmp
# (at most two counters of one block per pass: a third aborts rocprofv3 -- "exceeds the capabilities of the hardware")
groups=("TA_TA_BUSY_sum GRBM_GUI_ACTIVE"
        "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
        "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
        "TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
        "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
        "TD_TD_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"
        "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_CYCLES"
        "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAIT_INST_LDS"
        "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" )
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  ( cd /tmp && env "$@" timeout 240 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out" -o "m$i" -- python /root/repo/tools/run_c4_iterations.py > "$out/m$i.log" 2>&1 ) || echo "group '$g' failed"
done
python /root/repo/tools/pmc_summarize.py "$out" nn_ | cut -c1-60,150-400
