import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch.nn.utils.rnn import pack_sequence
from padertorch_amd.ops import packed_lstm
from padertorch_amd import _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
for B, T, H in [(64, 503, 600), (48, 253, 600)]:
    lstm = torch.nn.LSTM(257, H, 1, bidirectional=True).to(dev)
    xs = [torch.randn(T, 257, device=dev, requires_grad=True) for _ in range(B)]
    for it in range(3):
        p = pack_sequence(xs)
        _lib.KERNEL_TIMERS = []
        y = packed_lstm(lstm, p)
        y.data.sum().backward()
        torch.cuda.synchronize()
        tm = {}
        for n, a, b in _lib.KERNEL_TIMERS:
            tm[n] = tm.get(n, 0) + a.elapsed_time(b)
        _lib.KERNEL_TIMERS = None
    print(f'B={B} T={T} H={H}: fwd {tm["lstm_forward"]*1e3/T:.2f} us/step, bwd {tm["lstm_backward"]*1e3/T:.2f} us/step')
