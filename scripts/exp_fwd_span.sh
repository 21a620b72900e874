for cfg in "PTMI_LSTM_JT=12" "PTMI_LSTM_JT=16" "PTMI_LSTM_JT=20" "PTMI_LSTM_JT=20 PTMI_LSTM_FWD_SPAN=1" "PTMI_LSTM_JT=24" "PTMI_LSTM_JT=24 PTMI_LSTM_FWD_SPAN=1"; do
  echo "== $cfg"; env $cfg python scripts/exp_lstm.py 2>&1 | grep "B=32"
done
