#!/bin/bash
# Soak of the training step (gpurun -- bash scripts/soak.sh): cold starts of every single-GPU configuration (the first steps run
# unsynchronised: the situation in which a kernel that waits for undispatched workgroups next to a persistent recurrence would hang)
# and long runs; a recurrence watchdog timeout or a non-finite value ends bench.py with an error.  -> gpurun_out/soak.txt
out=gpurun_out/soak.txt; mkdir -p gpurun_out; : > $out
run() { local t0=$(date +%s.%N); timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('ok  %-28s %8.3f ms/step %9.0f frames/s' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['value']))
except Exception as e:
    print('FAILED', ' '.join(sys.argv[1:]), e)
" "$@" >> $out; }
for rep in 1 2 3; do
  for cfg in c2 c3 c5; do run --config $cfg --steps 8 --warmup 0; done
done
run --config c2 --steps 3000 --warmup 10
run --config c3 --steps 600 --warmup 5
run --config c4 --steps 100 --warmup 2
run --config c5 --steps 600 --warmup 5
run --config c2 --steps 500 --warmup 5 --bf16
run --config c2 --steps 300 --warmup 5 --sync-checks
# ragged batches (lengths U[3 s, 6 s], the per-pattern bookkeeping rebuilt every step): cold starts and long runs
for cfg in c2 c3 c5; do run --config $cfg --ragged --steps 8 --warmup 0; done
run --config c2 --ragged --steps 1500 --warmup 10
run --config c3 --ragged --steps 300 --warmup 5
run --config c5 --ragged --steps 300 --warmup 5
cat $out
