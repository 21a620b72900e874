#!/bin/bash
# usage: pmc_kernel.sh <tag> <kernel-name-filter> -- <command...>
# PMC passes (each its own rocprofv3 run); per-kernel mean counter values -> gpurun_out/p/pmc_<tag>.txt
tag=$1; flt=$2; shift 3
cd /tmp && export TMPDIR=/tmp
out=/root/repo/gpurun_out/p/pmc_$tag.txt
mkdir -p /root/repo/gpurun_out/p; rm -f $out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmc_$tag$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_$tag$i -o p -- "$@" > /tmp/pmc_$tag$i.log 2>&1 </dev/null
  db=$(find /tmp/pmc_$tag$i -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python /root/repo/scripts/pmc_summary.py "$db" "$flt" >> $out 2>&1 </dev/null; else echo "pass $i: no db" >> $out; tail -3 /tmp/pmc_$tag$i.log >> $out; fi
done
cat $out
