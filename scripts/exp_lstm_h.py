"""Forward / backward recurrence time per step for one (B, T, H) given as arguments: python scripts/exp_lstm_h.py 32 253 384"""
import sys, time, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch.nn.utils.rnn import pack_sequence
from padertorch_amd.ops import packed_lstm
from padertorch_amd import _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, T, H = (int(v) for v in sys.argv[1:4])
lstm = torch.nn.LSTM(257, H, 1, bidirectional=True).to(dev)
xs = [torch.randn(T, 257, device=dev, requires_grad=True) for _ in range(B)]
ref = None
for it in range(4):
    p = pack_sequence(xs)
    _lib.KERNEL_TIMERS = []
    y = packed_lstm(lstm, p)
    y.data.sum().backward()
    torch.cuda.synchronize()
    tm = {n: a.elapsed_time(b) for n, a, b in _lib.KERNEL_TIMERS}
    _lib.KERNEL_TIMERS = None
print(f'DBG {os.environ.get("PTMI_LSTM_DBG")} SPAN {os.environ.get("PTMI_LSTM_SPAN")} B={B} T={T} H={H}: fwd {tm["lstm_forward"]*1e3/T:.2f} us/step, '
      f'bwd {tm["lstm_backward"]*1e3/T:.2f} us/step, checksum {float(y.data.double().sum()):.6f} finite {bool(torch.isfinite(y.data).all())}')
