#!/bin/bash
# A/B runs of the CAPTURED step in one box: scripts/ab_graph.sh "<python statements>" ...  ("-" = defaults); CONFIG=c4 STEPS=20
for stmt in "$@"; do
  [ "$stmt" = "-" ] && stmt="pass"
  for rep in 1 2; do
    ms=$(python -c "
import sys
sys.argv = ['exp_graph_step.py', '--config', '${CONFIG:-c2}', '--replay-only', '${STEPS:-40}']
sys.path.insert(0, 'scripts')
import padertorch_amd.ops.lstm as L, padertorch_amd.ops.gemm as G
$stmt
import exp_graph_step
exp_graph_step.main()
" 2>/dev/null | tail -1)
    echo "$ms   [$stmt]"
  done
done
