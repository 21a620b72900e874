#!/bin/bash
# PMC passes over the iSTFT experiment (each pass = its own rocprofv3 run; summaries -> gpurun_out/p/)
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/p
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  timeout 150 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc$i -o p -- python /root/repo/scripts/exp_istft.py > /tmp/pmc$i.log 2>&1 </dev/null
  db=$(find /tmp/pmc$i -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python /root/repo/scripts/pmc_summary.py "$db" istft >> /root/repo/gpurun_out/p/pmc_istft.txt 2>&1 </dev/null; else echo "pass $i: no db" >> /root/repo/gpurun_out/p/pmc_istft.txt; tail -3 /tmp/pmc$i.log >> /root/repo/gpurun_out/p/pmc_istft.txt; fi
done
cat /root/repo/gpurun_out/p/pmc_istft.txt
