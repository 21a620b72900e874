import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch.nn.utils.rnn import pack_sequence
from padertorch_amd.ops import packed_lstm
from padertorch_amd import _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
import os
print('DBG', os.environ.get('PTMI_LSTM_DBG'))
BATCHES = [int(v) for v in sys.argv[sys.argv.index('--batches') + 1:]] if '--batches' in sys.argv else [32, 16, 1]
for B, T, H in [(B_, 253, 600) for B_ in BATCHES]:
    lstm = torch.nn.LSTM(257, H, 1, bidirectional=True).to(dev)
    xs = [torch.randn(T, 257, device=dev, requires_grad=True) for _ in range(B)]
    for it in range(3):
        p = pack_sequence(xs)
        _lib.KERNEL_TIMERS = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = packed_lstm(lstm, p)
        t1 = time.perf_counter()
        y.data.sum().backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        tm = {n: a.elapsed_time(b) for n, a, b in _lib.KERNEL_TIMERS}
        _lib.KERNEL_TIMERS = None
    print(f'B={B} T={T} H={H}: fwd {tm["lstm_forward"]*1e3/T:.2f} us/step, bwd {tm["lstm_backward"]*1e3/T:.2f} us/step, '
          f'host enqueue fwd {(t1-t0)*1e6/T:.2f} us/step')
