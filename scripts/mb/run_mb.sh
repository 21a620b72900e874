#!/bin/bash
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out/mb
O=../../gpurun_out/mb
{
for w in 0; do
 for c in 0; do
  timeout 60 ./xcd_chain_handoff 0 4 50 12 0 64 2000 $w $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 0 4 30 20 0 64 2000 $w $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 1 4 30 20 256 100 2000 $w $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 2 4 30 20 256 100 2000 $w $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 1 8 30 20 256 100 2000 $w $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 1 4 8 75 256 100 2000 $w $c 1 | tail -2
  timeout 60 ./xcd_chain_handoff 1 4 16 38 256 100 2000 $w $c 1 | tail -2
 done
done
} > $O/handoff2.txt 2>&1
cat $O/handoff2.txt
