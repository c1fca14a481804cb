#!/bin/bash
# Collect PMC counters for a command, ONE rocprofv3 pass per counter group
# (--pmc with --kernel-trace only, as the pool requires), CSV output under $1.
# usage: tools/pmc_collect.sh <outdir> -- <command...>
set -u
out=$1; shift; shift
mkdir -p "$out"
export TMPDIR=/tmp
groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"
        "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr"
        "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS")
for g in "${groups[@]}"; do
  tag=$(echo "$g" | tr ' ' '_' | cut -c1-20)
  ( cd /tmp && rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$out" -o "$tag" -- "$@" > "$out/$tag.log" 2>&1 ) || echo "group '$g' failed (see $out/$tag.log)"
done
