#!/bin/bash
# kernel trace of scripts/bench_configs.py-like single config: usage profile_config.sh <python-snippet-file> <tag>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$2
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_$2 -o p -- python $1 > /tmp/kt_$2.log 2>&1 </dev/null
db=$(find /tmp/kt_$2 -name "*.db" 2>/dev/null | head -1)
mkdir -p /root/repo/gpurun_out/p
if [ -n "$db" ]; then python /root/repo/scripts/profile_summary.py "$db" --top 30 > /root/repo/gpurun_out/p/kt_$2.txt 2>&1 </dev/null; else tail -5 /tmp/kt_$2.log > /root/repo/gpurun_out/p/kt_$2.txt; fi
cut -c1-180 /root/repo/gpurun_out/p/kt_$2.txt | head -34
