#!/bin/bash
# Round-6 soak on the GPU box: cold starts and long runs of the captured step in its three forms (single graph, two graphs around the
# data-parallel exchange under a one-rank RCCL group, one graph for ragged batches).  -> gpurun_out/r6_soak.txt
repo=$(pwd); out=$repo/gpurun_out/r6_soak.txt; : > $out
run() { r=$(timeout 900 python $repo/bench.py "$@" --no-cpu-baseline --no-extras --no-kernel-events 2>/tmp/soak.err | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%8.3f ms/step %9d frames/s' % (d['ms_per_step'], d['value']))" 2>/dev/null); if [ -n "$r" ]; then echo "ok  $*  $r" >> $out; else echo "FAIL $*: $(tail -2 /tmp/soak.err | tr '\n' ' ')" >> $out; fi; }
for i in 1 2 3; do for c in c2 c3 c5; do run --config $c --steps 8 --warmup 0; done; run --config c2 --dp-graph --steps 8 --warmup 0; done
run --config c2 --steps 3000 --warmup 10
run --config c3 --steps 600 --warmup 5
run --config c4 --steps 100 --warmup 2
run --config c5 --steps 600 --warmup 5
run --config c2 --dp-graph --steps 2000 --warmup 5
run --config c4 --dp-graph --steps 60 --warmup 2
timeout 900 python $repo/scripts/soak_ragged_graph.py >> $out 2>&1
cat $out
