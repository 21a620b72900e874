"""Masked (row-slot) vs uniform backward recurrence under the timing ablations (scripts/build_ablate.sh): where do the masked kernel's
+0.45 us per step come from?  Equal lengths in both (32 examples of 253 frames in 32 slots), so the data are the same."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch.nn.utils.rnn import PackedSequence, pack_sequence
from padertorch_amd import _lib
_lib.LIB_PATH = Path(__file__).resolve().parent / 'mb' / 'libptmi_ablate.so'
from padertorch_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
S, T, H = 32, 253, 600
layout = ops.sequence.SlotLayout.cached(tuple([T] * S), S, dev)
lstm = torch.nn.LSTM(257, H, 1, bidirectional=True).to(dev)
x = torch.randn(T * S, 257, device=dev, requires_grad=True)
xs = [torch.randn(T, 257, device=dev, requires_grad=True) for _ in range(S)]
for name, bits in [('as shipped', 0), ('no waiting', 8192), ('no cold loads', 4096), ('no plane / row-major stores', 2048), ('no hand-off stores', 16384),
                   ('no MFMAs', 64), ('operands not waited for', 128), ('no waiting, no cold, no stores, no MFMA, no operands', 8192 | 4096 | 2048 | 16384 | 64 | 128)]:
    os.environ['PTMI_LSTM_DBG'] = str(bits)
    out = []
    for masked in (False, True):
        best = None
        for it in range(4):
            _lib.KERNEL_TIMERS = []
            if masked:
                y = ops.packed_lstm(lstm, PackedSequence(x, torch.full((T,), S, dtype=torch.int64)), meta=layout.meta).data
            else:
                y = ops.packed_lstm(lstm, pack_sequence(xs)).data
            torch.nan_to_num(y).sum().backward()
            torch.cuda.synchronize()
            tm = {n: a.elapsed_time(b) for n, a, b in _lib.KERNEL_TIMERS}
            _lib.KERNEL_TIMERS = None
            cur = tm['lstm_backward'] * 1e3 / T
            best = cur if best is None else min(best, cur)
        out.append(best)
    print(f'{name:60s} uniform {out[0]:5.2f}   masked {out[1]:5.2f}   (+{out[1] - out[0]:.2f})', flush=True)
