# rocprofv3 kernel trace of a short bench run -> gpurun_out/timeline_all2.txt (every kernel of one step in start order per queue;
# the profiler inflates cross-queue gaps).  usage: bash scripts/timeline_all.sh [bench.py arguments, e.g. --config c3]
repo=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $repo/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras "$@" > /tmp/kt.log 2>&1 </dev/null
db=$(find /tmp/kt -name "*.db" | head -1)
python $repo/scripts/step_timeline.py "$db" --min-us 0 > $repo/gpurun_out/timeline_all2.txt 2>&1
tail -3 /tmp/kt.log
