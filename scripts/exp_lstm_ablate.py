"""Timing ablations of one recurrence step's terms on the REAL kernels (csrc/lstm_split.hip built with -DPTMI_LSTM_ABLATE:
scripts/build_ablate.sh -> scripts/mb/libptmi_ablate.so).  us per time step, H = 600, T = 253, B = 32 (and 16), one layer, stand-alone
launches; every variant but the first computes garbage.  -> profiles/r6_lstm_ablations.txt"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch.nn.utils.rnn import pack_sequence
from padertorch_amd import _lib
_lib.LIB_PATH = Path(__file__).resolve().parent / 'mb' / 'libptmi_ablate.so'
from padertorch_amd.ops import packed_lstm, lstm as L

dev = torch.device('cuda:0')
NOWAIT, LOADS, MFMA, COLD, PRE, STORES, HAND = 8192, 128, 64, 4096, 1024, 2048, 16384
VARIANTS = [
    ('as shipped', 0),
    ('no waiting (every operand accepted as it arrives)', NOWAIT),
    ('no waiting, operand values not waited for', NOWAIT | LOADS),
    ('no waiting, no MFMAs', NOWAIT | MFMA),
    ('no waiting, no operands, no MFMAs', NOWAIT | LOADS | MFMA),
    ('no waiting, no cold loads (saved activations / next pre-activations)', NOWAIT | COLD | PRE),
    ('no waiting, no row-major / plane stores', NOWAIT | STORES),
    ('no waiting, no hand-off stores', NOWAIT | HAND),
    ('no waiting, no stores at all', NOWAIT | STORES | HAND),
    ('no waiting, nothing but reduction + barrier + gate arithmetic', NOWAIT | LOADS | MFMA | COLD | PRE | STORES | HAND),
    ('  ... and no clock read', NOWAIT | LOADS | MFMA | COLD | PRE | STORES | HAND | 32768),
    ('  ... and no LDS reduction / barrier', NOWAIT | LOADS | MFMA | COLD | PRE | STORES | HAND | 32768 | 65536),
    ('  ... and no transcendentals (forward)', NOWAIT | LOADS | MFMA | COLD | PRE | STORES | HAND | 32768 | 65536 | 131072),
    ('waiting, no cold loads', COLD | PRE),
    ('waiting, no row-major / plane stores', STORES),
]
for B in (32,):
    torch.manual_seed(0)
    T, H = 253, 600
    lstm = torch.nn.LSTM(257, H, 1, bidirectional=True).to(dev)
    xs = [torch.randn(T, 257, device=dev, requires_grad=True) for _ in range(B)]
    print(f'# B = {B}, T = {T}, H = {H}: us per time step (forward / backward)')
    for name, bits in VARIANTS:
        os.environ['PTMI_LSTM_DBG'] = str(bits)
        keep = L.CHECK_PERSISTENT_ERRORS
        best = None
        for it in range(4):
            p = pack_sequence(xs)
            _lib.KERNEL_TIMERS = []
            y = packed_lstm(lstm, p)
            torch.nan_to_num(y.data).sum().backward()
            torch.cuda.synchronize()
            tm = {n: a.elapsed_time(b) for n, a, b in _lib.KERNEL_TIMERS}
            _lib.KERNEL_TIMERS = None
            cur = (tm['lstm_forward'] * 1e3 / T, tm['lstm_backward'] * 1e3 / T)
            best = cur if best is None else (min(best[0], cur[0]), min(best[1], cur[1]))
        print(f'{name:75s} dbg={bits:6d}  {best[0]:5.2f} / {best[1]:5.2f}', flush=True)
