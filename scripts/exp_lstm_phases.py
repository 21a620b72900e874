"""Per-phase clock sums of the instrumented persistent forward kernel (PTMI_LSTM_PHASES=1), through the C ABI."""
import os, sys
from pathlib import Path
os.environ['PTMI_LSTM_PHASES'] = '1'
os.environ['PTMI_LSTM_DAF'] = '0'          # the instrumented kernel is the flag-protocol one
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from padertorch_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
B, T, H, ndir = 32, 253, 600, 2
KP, G, rows = (H + 15) // 16 * 16, 4 * H, B * T
torch.manual_seed(0)
gates0 = torch.randn(rows, ndir * G, device=dev) * 0.5
w = (torch.randn(ndir, G, KP, device=dev) * 0.05)
w[:, :, H:] = 0
bs = torch.full((T,), B, dtype=torch.int32, device=dev)
offs = (torch.arange(T, dtype=torch.int64, device=dev) * B)
n = int(lib.ptmi_lstm_scratch_elems(T, ndir, B, H, 0))
names = ['top (loop head, bookkeeping)', 'poll', 'barrier1', 'issue operand + prefetch loads', 'wait operands + mfma + lds write', 'barrier2',
         'lds reduce + activations', 'hand-off stores issued', 'drain', 'barrier3', 'flag + trailing stores', 'loop back edge (after the trailing stores)']
for it in range(3):
    gates = gates0.clone()
    hy = torch.empty(rows, ndir * H, device=dev); c = torch.empty_like(hy)
    scratch = torch.empty(n, dtype=torch.int32, device=dev)
    rc = lib.ptmi_lstm_forward_persistent(gates.data_ptr(), hy.data_ptr(), c.data_ptr(), None, w.data_ptr(), None, bs.data_ptr(),
                                          offs.data_ptr(), scratch.data_ptr(), T, B, rows, H, KP, ndir, 0, None, _lib.stream(dev))
    torch.cuda.synchronize()
    assert rc == 0, rc
ph = scratch[:24].view(torch.int64).cpu().numpy().astype(np.float64) * 10.0 / (T - 1)      # ns per step (100 MHz clock)
for nme, v in zip(names, ph):
    print(f'{nme:24s} {v:8.1f} ns')
print(f'{"sum":24s} {ph.sum():8.1f} ns')
