#!/bin/bash
# Runs the micro-benchmarks of DESIGN.md section 3.3 / 3.9 on the GPU box (gpurun -- bash scripts/mb/run_mb.sh).
# Build first (here, no GPU needed):  cd scripts/mb && for f in *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 $f -o ${f%.hip}; done
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out/mb
O=../../gpurun_out/mb
./split_mfma_accuracy > $O/accuracy.txt 2>&1
./xcd_dispatch_probe > $O/dispatch_probe.txt 2>&1
{ timeout 120 ./gemm_planes; timeout 120 ./gemm_planes 8096 1200 4800; timeout 120 ./gemm_planes 8192 8192 4096; } > $O/gemm_planes.txt 2>&1
{
for c in 1 0; do
  # all-gather (forward form): spread over the XCDs (round-1 protocol), one XCD with sc1 / plain loads, 8 chains
  timeout 60 ./xcd_chain_handoff 0 4 50 12 0 64 1000 0 $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 0 4 30 20 0 64 1000 0 $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 1 4 30 20 256 100 1000 0 $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 2 4 30 20 256 100 1000 0 $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 3 4 30 20 256 100 1000 0 $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 3 8 30 20 256 100 1000 0 $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 1 4 50 12 512 64 1000 0 $c 1 | tail -2
  # reduce-scatter (transposed backward form)
  timeout 60 ./xcd_chain_handoff 4 4 30 20 256 100 1000 0 $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 5 4 30 20 0 100 1000 0 $c 1 | tail -2
done
} > $O/handoff.txt 2>&1
tail -4 $O/accuracy.txt; cat $O/dispatch_probe.txt; cat $O/handoff.txt
