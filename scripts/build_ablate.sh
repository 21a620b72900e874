#!/bin/bash
# libptmi with the recurrence kernels' timing ablations compiled in (-DPTMI_LSTM_ABLATE: csrc/lstm_split.hip) -> scripts/mb/libptmi_ablate.so
# (the product library compiles none of them: identical device code with and without the hooks in the source).
repo=$(cd "$(dirname "$0")/.." && pwd)
objs=""
mkdir -p /tmp/ablate_obj
for f in $repo/padertorch_amd/csrc/*.hip; do
  o=/tmp/ablate_obj/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DPTMI_LSTM_ABLATE -I$repo/include -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o $repo/scripts/mb/libptmi_ablate.so && ls -la $repo/scripts/mb/libptmi_ablate.so
