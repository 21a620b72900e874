#!/bin/bash
# A/B runs of the bench step under environment knobs: scripts/ab.sh "VAR=1" "VAR2=0 VAR3=1" ...   ("-" = defaults)
for env in "$@"; do
  [ "$env" = "-" ] && env=""
  for rep in 1 2; do
    ms=$(env $env python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "ms/step $ms   [$env]"
  done
done
