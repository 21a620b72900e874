repo=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kth; timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/kth -o p -- python $repo/scripts/exp_graph_step.py --replay-only 12 "$@" > /tmp/kth.log 2>&1 </dev/null
db=$(find /tmp/kth -name "*.db" | head -1)
python $repo/scripts/timeline_head.py "$db" > $repo/gpurun_out/timeline_head.txt 2>&1
tail -2 /tmp/kth.log; cat $repo/gpurun_out/timeline_head.txt | cut -c1-200
