# kernel durations of the stand-alone recurrences: row slots against equal lengths of the same T (rocprofv3 kernel trace)
repo=$(pwd); mkdir -p $repo/gpurun_out/p; cd /tmp && export TMPDIR=/tmp
for v in slots uni; do
  if [ $v = slots ]; then cmd="python $repo/scripts/exp_lstm_slots.py 64 32"; else cmd="python $repo/scripts/exp_lstm_h.py 32 577 600"; fi
  rm -rf /tmp/kl_$v; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kl_$v -o p -- $cmd > /tmp/kl_$v.log 2>&1 </dev/null
  db=$(find /tmp/kl_$v -name "*.db" | head -1)
  echo "== $v"; python $repo/scripts/profile_summary.py "$db" --top 8 2>&1 | cut -c1-170
done
