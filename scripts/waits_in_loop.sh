#!/bin/bash
# Vector-memory operations and wait counts of a kernel's time loop, from the device assembly (no GPU needed):
#   scripts/waits_in_loop.sh <csrc file stem> <mangled-name fragment> [first|last loop header]
# e.g. scripts/waits_in_loop.sh lstm_split lstm_fwd_daf_kernelILi12ELi8ELi3ELi1ELi4EEE last
#      scripts/waits_in_loop.sh lstm_split lstm_bwd_split_kernelILi8ELi10ELi1ELi3ELi16ELb1ELb1ELb1ELb0EEE first
# What to look for: flat_load / flat_store (a FLAT access makes every later wait vmcnt(0) lgkmcnt(0)), s_waitcnt vmcnt(0) between the
# workgroup barrier and the hand-off store or at the loop head, global_load behind an s_and_saveexec (a conditional load: its use waits
# with vmcnt(0)).
stem=$1; frag=$2; which=${3:-first}
repo=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/waits_$stem.s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc --cuda-device-only -S -o $out -I$repo/include $repo/padertorch_amd/csrc/$stem.hip 2>/dev/null || exit 1
a=$(grep -n "^_ZN4ptmi[0-9]*${frag}.*:" $out | head -1 | cut -d: -f1)
[ -z "$a" ] && { echo "no kernel matches $frag"; grep -o "^_ZN4ptmi[^:]*" $out | head -40; exit 1; }
e=$(awk -v a=$a 'NR>a && /^\.Lfunc_end/ {print NR; exit}' $out)
sed -n "${a},${e}p" $out > /tmp/waits_kernel.s
if [ $which = last ]; then l=$(grep -n "Loop Header: Depth=1" /tmp/waits_kernel.s | tail -1 | cut -d: -f1); else l=$(grep -n "Loop Header: Depth=1" /tmp/waits_kernel.s | head -1 | cut -d: -f1); fi
echo "# $(sed -n 1p /tmp/waits_kernel.s | cut -c1-120)"
echo "# flat accesses in the kernel: $(grep -c 'flat_load\|flat_store' /tmp/waits_kernel.s)"
awk -v l=$l 'NR>=l' /tmp/waits_kernel.s | grep -n "s_waitcnt\|global_load\|buffer_load\|flat_\|global_store\|buffer_store\|s_barrier\|s_memrealtime\|saveexec\|s_endpgm\|Loop Header" | awk '{print $1, $2, $3, $4}'
