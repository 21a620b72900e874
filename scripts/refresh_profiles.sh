#!/bin/bash
# Refresh the round's measurements on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel trace of the same bench command, FETCH_SIZE / WRITE_SIZE PMC passes
#   (separate runs, kernel-trace only), kernel micro-benchmarks.  Text summaries -> gpurun_out/p/.
# usage: scripts/refresh_profiles.sh <round-tag>
tag=${1:-r5}
repo=$(pwd)
out=$repo/gpurun_out/p
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 1200 python $repo/bench.py 2>/dev/null | tail -1 > $out/${tag}_bench.json
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /tmp/kt.log 2>&1 </dev/null
db=$(find /tmp/kt -name "*.db" 2>/dev/null | head -1)
if [ -n "$db" ]; then python $repo/scripts/profile_summary.py "$db" --top 40 > $out/${tag}_kernel_trace_bench.txt 2>&1 </dev/null; else tail -5 /tmp/kt.log > $out/${tag}_kernel_trace_bench.txt; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc_$c.log 2>&1 </dev/null
  db=$(find /tmp/pmc_$c -name "*.db" 2>/dev/null | head -1)
  lc=$(echo $c | tr A-Z a-z)
  if [ -n "$db" ]; then python $repo/scripts/profile_summary.py "$db" --pmc > $out/${tag}_pmc_$lc.txt 2>&1 </dev/null; else tail -5 /tmp/pmc_$c.log > $out/${tag}_pmc_$lc.txt; fi
done
timeout 600 python $repo/scripts/bench_kernels.py --big 2>/dev/null | grep '^{' > $out/${tag}_kernel_microbench.jsonl
( timeout 600 python $repo/scripts/bench_gemm.py 2>/dev/null | grep '^{'; timeout 600 python $repo/scripts/bench_gemm.py 32192 2>/dev/null | grep '^{' ) > $out/${tag}_gemm_microbench.jsonl
# one step's kernels in start order (critical path) from the kernel trace
db=$(find /tmp/kt -name "*.db" 2>/dev/null | head -1)
[ -n "$db" ] && python $repo/scripts/step_timeline.py "$db" --min-us 25 > $out/${tag}_step_timeline.txt 2>&1 </dev/null
# the other configurations and modes: one JSON line each
( for cfg in c3 c4 c5; do timeout 900 python $repo/bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1; done
  timeout 900 python $repo/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --eager 2>/dev/null | tail -1
  timeout 900 python $repo/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --bf16 2>/dev/null | tail -1 ) > $out/${tag}_configs.jsonl
# the kernels of the TIMED mode (replays): what bench.py's frac_replay_profile reads
for cfg in c2 c3; do
  rm -rf /tmp/ktr; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ktr -o p -- python $repo/bench.py --config $cfg --no-extras --no-kernel-events --no-cpu-baseline > /tmp/ktr.log 2>&1 </dev/null
  db=$(find /tmp/ktr -name "*.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python $repo/scripts/profile_summary.py "$db" --top 40 > $out/${tag}_kernel_trace_replay_$cfg.txt 2>&1 </dev/null
done
# the captured step's schedule: rocprofv3 kernel trace of graph replays (a replay does not depend on the host), c2 and c3
cd $repo; bash scripts/timeline_graph.sh > /dev/null 2>&1; cp gpurun_out/timeline_graph.txt $out/${tag}_graph_timeline.txt
bash scripts/timeline_graph.sh --config c3 > /dev/null 2>&1; cp gpurun_out/timeline_graph.txt $out/${tag}_graph_timeline_c3.txt; cd /tmp
# where the EAGER step's critical path goes without a profiler attached (HIP events around the phase-marking launches)
timeout 300 python $repo/scripts/phase_events.py 2>/dev/null | grep -E '^#|^[ms] ' > $out/${tag}_phase_events.txt
[ -x $repo/scripts/mb/pack_t ] && { cd $repo/scripts/mb; { ./pack_t 8096 2400 4800; ./pack_t 32192 2400 4800; ./pack_t 32192 1200 1200; } > $out/${tag}_mb_pack_t.txt 2>&1; cd /tmp; }
# ragged batches (lengths U[3 s, 6 s], bookkeeping rebuilt every step): a reported mode
( for a in "" "--config c3" "--config c5"; do timeout 600 python $repo/bench.py $a --ragged --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ragged $a: %.2f ms/step, %d frames/s, %d frames/step' % (d['ms_per_step'], d['value'], d['config']['frames_per_step']))"; done
  timeout 300 python $repo/scripts/phase_events.py --ragged 2>/dev/null | grep -E '^#|^[ms] ' ) > $out/${tag}_ragged.txt
timeout 600 python $repo/scripts/bf16_delta.py 2>/dev/null | tail -1 > $out/${tag}_bf16_delta.json
timeout 300 python $repo/scripts/exp_lstm.py 2>/dev/null | grep 'B=' > $out/${tag}_lstm_us_per_step.txt
ls -la $out
