"""Experiment (round 6): forward recurrence with chains on ONE XCD (20-unit tiles: 30 workgroups per chain), plain hand-off stores, and the
bookkeeping decided at compile time - alone and together.  us per time step, H = 600, T = 253, B = 32; checks the output against the
shipped configuration."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch.nn.utils.rnn import pack_sequence
lib = sys.argv[1] if len(sys.argv) > 1 else 'libptmi_exp.so'
from padertorch_amd import _lib
_lib.LIB_PATH = Path(__file__).resolve().parent / 'mb' / lib
from padertorch_amd.ops import packed_lstm, lstm as L
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, T, H = 32, 253, 600
lstm = torch.nn.LSTM(257, H, 1, bidirectional=True).to(dev)
xs = [torch.randn(T, 257, device=dev) for _ in range(B)]
ref = None
PLAIN = 1 << 25
for name, env in [('shipped (12-unit tiles, 2 XCDs per chain)', {}),
                  ('20-unit tiles, 2 XCDs per chain', dict(PTMI_LSTM_JT='20')),
                  ('20-unit tiles, 1 XCD per chain', dict(PTMI_LSTM_JT='20', PTMI_LSTM_SPAN='1')),
                  ('20-unit tiles, 1 XCD per chain, plain stores', dict(PTMI_LSTM_JT='20', PTMI_LSTM_SPAN='1', PTMI_LSTM_DBG=str(PLAIN))),
                  ('20-unit tiles, 1 XCD, plain stores, fixed hold 40', dict(PTMI_LSTM_JT='20', PTMI_LSTM_SPAN='1', PTMI_LSTM_DBG=str(PLAIN | (10 << 16) | (1 << 24)))),
                  ('20-unit tiles, 1 XCD, plain stores, fixed hold 20', dict(PTMI_LSTM_JT='20', PTMI_LSTM_SPAN='1', PTMI_LSTM_DBG=str(PLAIN | (5 << 16) | (1 << 24)))),
                  ('20-unit tiles, 1 XCD, plain stores, fixed hold 60', dict(PTMI_LSTM_JT='20', PTMI_LSTM_SPAN='1', PTMI_LSTM_DBG=str(PLAIN | (15 << 16) | (1 << 24))))]:
    for k in ('PTMI_LSTM_JT', 'PTMI_LSTM_SPAN', 'PTMI_LSTM_DBG'):
        os.environ.pop(k, None)
    os.environ.update(env)
    best = None
    L.CHECK_PERSISTENT_ERRORS = True
    try:
        with torch.no_grad():
            for it in range(5):
                _lib.KERNEL_TIMERS = []
                y = packed_lstm(lstm, pack_sequence(xs)).data
                torch.cuda.synchronize()
                tm = {n: a.elapsed_time(b) for n, a, b in _lib.KERNEL_TIMERS}
                _lib.KERNEL_TIMERS = None
                cur = tm['lstm_forward'] * 1e3 / T
                best = cur if best is None else min(best, cur)
        if ref is None:
            ref = y.clone()
        print(f'{lib:22s} {name:55s} {best:5.2f} us/step   max |y - shipped| = {float((y - ref).abs().max()):.1e}', flush=True)
    except Exception as e:
        print(f'{lib:22s} {name:55s} FAILED: {str(e)[:100]}', flush=True)
