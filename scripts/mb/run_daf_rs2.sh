#!/bin/bash
# second scan: longer holds (the first scan's optimum lay beyond its range)
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out/mb
O=../../gpurun_out/mb/daf_rs2.txt
{
echo "# all-gather, span 2, P 38"
for h in 125 140 155 170 185 200 220 250 300; do timeout 120 ./daf_rs 0 $h 253 0 | tail -1; done
echo "# reduce-scatter, span 2, P 38"
for h in 140 155 170 185 200 220 250 300; do timeout 120 ./daf_rs 1 $h 253 0 | tail -1; done
echo "# reduce-scatter, ONE XCD per chain (P 32), write-through"
for h in 120 140 155 170 185 200 250; do timeout 120 ./daf_rs 1 $h 253 0 1 32 0 | tail -1; done
echo "# reduce-scatter, span 2, P 32"
for h in 120 140 155 170 185 200 250; do timeout 120 ./daf_rs 1 $h 253 0 2 32 0 | tail -1; done
echo "# phases at the long holds"
timeout 120 ./daf_rs 0 200 253 0 2 38 0 1 | tail -3
timeout 120 ./daf_rs 1 200 253 0 2 38 0 1 | tail -3
} > $O 2>&1
cat $O
