export PTMI_LSTM_MAX_POLLS=20000
run() { timeout 120 python scripts/exp_lstm_h.py 32 253 384 2>&1 | tail -1; }
for sc0 in 1; do for h in 20 40 60 80; do
  d=$(( (h/4)<<16 | (1<<24) | (1<<25) | (sc0<<26) ))
  echo "== span 1 plain stores sc0 loads $sc0 hold $h"; PTMI_LSTM_SPAN=1 PTMI_LSTM_DBG=$d run
done; done
# ablations at span 1 / plain stores / hold 20: no waiting at all (8192), no MFMA? 
for extra in 8192; do d=$(( (20/4)<<16 | (1<<24) | (1<<25) | extra )); echo "== span 1 plain hold 20 + dbg $extra"; PTMI_LSTM_SPAN=1 PTMI_LSTM_DBG=$d run; done
