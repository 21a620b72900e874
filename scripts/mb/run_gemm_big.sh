#!/bin/bash
# gpurun -- bash scripts/mb/run_gemm_big.sh [shapes...]: the big-tile GEMM prototype at the step's shapes (build scripts/mb/gemm_big first)
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out/mb
O=../../gpurun_out/mb/gemm_big.txt
: > $O
if [ $# -gt 0 ]; then shapes=("$@"); else shapes=("8096 4800 1216" "8096 1200 4864" "32192 4800 1216" "8192 8192 4096" "8096 4800 288"); fi
for shape in "${shapes[@]}"; do
  timeout 300 ./gemm_big $shape >> $O 2>&1
done
cat $O
