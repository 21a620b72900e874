"""Recurrence time per step on the row-slot layout (masked kernels), stand-alone: python scripts/exp_lstm_slots.py [examples] [slots]"""
import sys, random
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch.nn.utils.rnn import PackedSequence
from padertorch_amd import ops, _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
n_ex = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rnd = random.Random(4321)
lens = sorted((rnd.randint(190, 378) for _ in range(n_ex)), reverse=True)
if len(sys.argv) > 3:          # equal lengths: the masked kernels without a sequence boundary inside the grid
    lens = [int(sys.argv[3])] * n_ex
layout = ops.sequence.SlotLayout.cached(tuple(lens), S, dev)
T, H = layout.T, 600
lstm = torch.nn.LSTM(257, H, 1, bidirectional=True).to(dev)
x = torch.randn(T * S, 257, device=dev, requires_grad=True)
for it in range(4):
    _lib.KERNEL_TIMERS = []
    y = ops.packed_lstm(lstm, PackedSequence(x, torch.full((T,), S, dtype=torch.int64)), meta=layout.meta).data
    y.sum().backward()
    torch.cuda.synchronize()
    tm = {n: a.elapsed_time(b) for n, a, b in _lib.KERNEL_TIMERS}
    _lib.KERNEL_TIMERS = None
print(f'examples {n_ex} slots {S} T {T} occupancy {layout.occupancy:.3f}: fwd {tm["lstm_forward"]*1e3/T:.2f} us/step, bwd {tm["lstm_backward"]*1e3/T:.2f} us/step '
      f'finite {bool(torch.isfinite(y).all())}')
