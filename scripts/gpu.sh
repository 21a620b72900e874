#!/bin/bash
# usage: scripts/gpu.sh <tag> <timeout-s> '<command>'   -> runs on the GPU box, log in gpurun_out/<tag>.log
tag=$1; to=$2; shift 2
/usr/local/graft/bin/gpurun --timeout $to -- "mkdir -p gpurun_out; ( $* ) > gpurun_out/$tag.log 2>&1; echo rc=\$? >> gpurun_out/$tag.log" > gpurun_out/${tag}_call.log 2>&1
tail -5 gpurun_out/${tag}_call.log | cut -c1-300
