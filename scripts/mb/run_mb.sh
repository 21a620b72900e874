#!/bin/bash
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out/mb
O=../../gpurun_out/mb
{
for c in 1 0; do
  # all-gather (forward form)
  timeout 60 ./xcd_chain_handoff 0 4 50 12 0 64 1000 0 $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 1 4 30 20 256 100 1000 0 $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 3 4 30 20 256 100 1000 0 $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 3 8 30 20 256 100 1000 0 $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 3 8 30 20 256 100 1000 0 $c 0 | tail -1
  # reduce-scatter (backward form): P * 16 * JT <= 10240 floats per consumer
  timeout 60 ./xcd_chain_handoff 4 4 30 20 256 100 1000 0 $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 4 8 30 20 256 100 1000 0 $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 4 8 30 20 256 100 1000 0 $c 0 | tail -1
  timeout 60 ./xcd_chain_handoff 5 4 30 20 0 100 1000 0 $c 1 | tail -2
done
} > $O/handoff3.txt 2>&1
cat $O/handoff3.txt
