#!/bin/bash
# A/B runs of the bench step in ONE box: scripts/ab.sh "<python statements>" ...   ("-" = defaults), two repetitions each, e.g.
#   scripts/ab.sh - "L.WGRAD_FROM_HANDOFF=False" "G.PRODUCTS=1"
# (L = padertorch_amd.ops.lstm, G = padertorch_amd.ops.gemm; BENCH_ARGS='--config c3' STEPS=30 for another configuration).  Differences below ~0.05 ms are noise, boxes differ by ~1-2 %.
for stmt in "$@"; do
  [ "$stmt" = "-" ] && stmt="pass"
  for rep in 1 2; do
    ms=$(python -c "
import sys
sys.argv = ['bench.py', '--steps', '${STEPS:-100}', '--warmup', '10', '--no-cpu-baseline', '--no-extras', '--eager'] + '${BENCH_ARGS:-}'.split()
import padertorch_amd.ops.lstm as L, padertorch_amd.ops.gemm as G
$stmt
import bench
bench.main()
" 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "ms/step $ms   [$stmt]"
  done
done
