# kernel statistics of the row-slot step next to the fixed-length step (rocprofv3 kernel trace of short bench runs) -> gpurun_out/p/
repo=$(pwd); mkdir -p $repo/gpurun_out/p; cd /tmp && export TMPDIR=/tmp
for v in fixed slots; do
  a=""; [ $v = slots ] && a="--ragged --row-slots"
  rm -rf /tmp/kt_$v; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o p -- python $repo/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras $a > /tmp/kt_$v.log 2>&1 </dev/null
  db=$(find /tmp/kt_$v -name "*.db" | head -1)
  python $repo/scripts/profile_summary.py "$db" --top 45 > $repo/gpurun_out/p/s4_kernels_$v.txt 2>&1
  tail -1 /tmp/kt_$v.log | cut -c1-200
done
