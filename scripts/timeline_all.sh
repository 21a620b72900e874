repo=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /tmp/kt.log 2>&1 </dev/null
db=$(find /tmp/kt -name "*.db" | head -1)
python $repo/scripts/step_timeline.py "$db" --min-us 0 > $repo/gpurun_out/timeline_all2.txt 2>&1
tail -3 /tmp/kt.log
