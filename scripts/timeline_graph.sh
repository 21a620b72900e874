# rocprofv3 kernel trace of graph replays (train.graphed.GraphedStep) -> gpurun_out/timeline_graph.txt; a replay does not depend on the
# host, so the profiler's slow-down of the launch path does not distort it.  usage: bash scripts/timeline_graph.sh [--config c3]
repo=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktg; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ktg -o p -- python $repo/scripts/exp_graph_step.py --replay-only 12 "$@" > /tmp/ktg.log 2>&1 </dev/null
db=$(find /tmp/ktg -name "*.db" | head -1)
python $repo/scripts/step_timeline.py "$db" --min-us 0 > $repo/gpurun_out/timeline_graph.txt 2>&1
tail -2 /tmp/ktg.log
