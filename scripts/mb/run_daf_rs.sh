#!/bin/bash
# gpurun -- bash scripts/mb/run_daf_rs.sh : the backward hand-off as all-gather (mode 0) against reduce-scatter (mode 1), data-as-flag
# protocol, hold scan; span 2 = the kernels' placement (a chain on two XCDs), span 1 = a chain on ONE XCD (P = 32) with write-through
# and with plain stores.  Build first: cd scripts/mb && hipcc --offload-arch=gfx950 -O3 -std=c++17 daf_rs.hip -o daf_rs
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out/mb
O=../../gpurun_out/mb/daf_rs.txt
{
echo "# correctness (every consumed word checked)"
timeout 120 ./daf_rs 0 100 253 1 | tail -2
timeout 120 ./daf_rs 1 100 253 1 | tail -2
timeout 120 ./daf_rs 1 60 253 1 1 32 1 | tail -2
echo "# all-gather, span 2, P 38: hold scan"
for h in 60 80 92 100 110 125; do timeout 120 ./daf_rs 0 $h 253 0 | tail -1; done
echo "# reduce-scatter, span 2, P 38: hold scan"
for h in 20 40 60 80 100 120 140 170; do timeout 120 ./daf_rs 1 $h 253 0 | tail -1; done
echo "# reduce-scatter, ONE XCD per chain (P 32), write-through stores"
for h in 20 40 60 80 100 140; do timeout 120 ./daf_rs 1 $h 253 0 1 32 0 | tail -1; done
echo "# reduce-scatter, ONE XCD per chain (P 32), plain stores (stay in the XCD's L2)"
for h in 0 10 20 40 60 80 100; do timeout 120 ./daf_rs 1 $h 253 0 1 32 1 | tail -1; done
echo "# reduce-scatter, span 2, P 32 (for comparison with the one-XCD rows)"
for h in 40 60 80 100 140; do timeout 120 ./daf_rs 1 $h 253 0 2 32 0 | tail -1; done
echo "# phases (thread 0 of workgroup 1 of chain 0, ns per step)"
timeout 120 ./daf_rs 0 100 253 0 2 38 0 1 | tail -3
timeout 120 ./daf_rs 1 80 253 0 2 38 0 1 | tail -3
timeout 120 ./daf_rs 1 20 253 0 1 32 1 1 | tail -3
} > $O 2>&1
cat $O
