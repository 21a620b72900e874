"""What follows a forward recurrence: the queue time between the end of packed_lstm (3 x BLSTM-600, B = 32, T = 253) and the end of a
tiny kernel enqueued right behind it, against the same tiny kernel behind a GEMM (a 75 us bubble sits behind the top layer's
recurrence in the training step: scripts/phase_events.py)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from torch.nn.utils.rnn import pack_sequence  # noqa: E402

import padertorch_amd as pt  # noqa: E402,F401
from padertorch_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
lstm = torch.nn.LSTM(257, 600, 3, bidirectional=True).to(dev)
x = pack_sequence([torch.randn(253, 257, device=dev) for _ in range(32)])
probe = torch.zeros(256, device=dev)


def gap(fn, follow):
    out = []
    for _ in range(12):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        r = fn()
        e1.record()
        follow(r)
        e2.record()
        torch.cuda.synchronize()
        out.append((e0.elapsed_time(e1) * 1e3, e1.elapsed_time(e2) * 1e3))
    out = out[4:]
    return sum(a for a, _ in out) / len(out), sum(b for _, b in out) / len(out)


with torch.no_grad():
    print('no grad : lstm %.1f us, then tiny kernel %.1f us' % gap(lambda: ops.packed_lstm(lstm, x), lambda r: probe.add_(1.)))
for p in lstm.parameters():
    p.grad = torch.zeros_like(p)
print('grad    : lstm %.1f us, then tiny kernel %.1f us' % gap(lambda: ops.packed_lstm(lstm, x), lambda r: probe.add_(1.)))
w = torch.randn(1200, 1200, device=dev)
print('grad    : lstm %.1f us, then torch.mm %.1f us' % gap(lambda: ops.packed_lstm(lstm, x), lambda r: torch.mm(r.data, w)))
a = torch.randn(8096, 1200, device=dev)
print('torch.mm %.1f us, then tiny kernel %.1f us' % gap(lambda: torch.mm(a, w), lambda r: probe.add_(1.)))
print('tiny %.1f us, then tiny kernel %.1f us' % gap(lambda: probe.add_(1.), lambda r: probe.add_(1.)))
