"""Random (B, T, H, I, layers, directions, ragged / equal lengths, initial state) configurations of
ops.packed_lstm against torch.nn.LSTM on the CPU: outputs, final states, input and parameter gradients.

    python scripts/fuzz_lstm.py [n=40] [seed=0]
"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from torch.nn.utils.rnn import pack_sequence
from padertorch_amd.ops import packed_lstm, lstm as L

dev = 'cuda:0'


def one(it, rng, verbose):
    H = int(rng.choice([4, 8, 24, 40, 100, 600, 624, 1024]))    # 624: 16-wavefront backward; 1024: one launch per step
    B = int(rng.integers(1, 72 if H < 600 else 40))
    T = int(rng.integers(1, 40 if H < 600 else 12))
    I = int(rng.integers(1, 40))
    layers = int(rng.integers(1, 4))
    bidir = bool(rng.integers(0, 2))
    equal = bool(rng.integers(0, 2))
    state = bool(rng.integers(0, 2))
    lens = [T] * B if equal else sorted((int(x) for x in rng.integers(1, T + 1, B)), reverse=True)
    lens[0] = T
    torch.manual_seed(it)
    ref = torch.nn.LSTM(I, H, layers, bidirectional=bidir)
    dut = torch.nn.LSTM(I, H, layers, bidirectional=bidir)
    dut.load_state_dict(ref.state_dict())
    dut = dut.to(dev)
    nd = 2 if bidir else 1
    xs = [torch.randn(l, I) for l in lens]
    xr = [x.clone().requires_grad_(True) for x in xs]
    xd = [x.clone().to(dev).requires_grad_(True) for x in xs]
    if state:
        hx = (torch.randn(layers * nd, B, H), torch.randn(layers * nd, B, H))
        yr, (hr, cr) = ref(pack_sequence(xr), hx)
        yd, (hd, cd) = packed_lstm(dut, pack_sequence(xd), hx=(hx[0].to(dev), hx[1].to(dev)))
    else:
        yr, (hr, cr) = ref(pack_sequence(xr))
        yd, (hd, cd) = packed_lstm(dut, pack_sequence(xd), return_state=True)
    g = torch.randn(yr.data.shape)
    (yr.data * g).sum().backward()
    (yd.data * g.to(dev)).sum().backward()
    errs = [float((yd.data.detach().cpu() - yr.data.detach()).abs().max()),
            float((hd.detach().cpu() - hr.detach()).abs().max()), float((cd.detach().cpu() - cr.detach()).abs().max())]
    errs += [float((a.grad.cpu() - b.grad).abs().max()) for a, b in zip(xd, xr)]
    gerr = max(float((pd.grad.cpu() - pr.grad).abs().max() / max(1.0, float(pr.grad.abs().max())))
               for pd, pr in zip(dut.parameters(), ref.parameters()))
    e = max(max(errs), gerr)
    if verbose:
        print(f'{it:3d} B={B:3d} T={T:3d} H={H:4d} I={I:3d} L={layers} bidir={int(bidir)} equal={int(equal)} '
              f'state={int(state)} max err {max(errs):.2e} param-grad rel {gerr:.2e}' + ('' if e < 5e-5 else '   <-- LARGE'),
              flush=True)
    return e


def run(n, seed, verbose=True):
    before = L.CHECK_PERSISTENT_ERRORS
    L.CHECK_PERSISTENT_ERRORS = True
    try:
        rng = np.random.default_rng(seed)
        return max(one(it, rng, verbose) for it in range(n))
    finally:
        L.CHECK_PERSISTENT_ERRORS = before


if __name__ == '__main__':
    print('worst', run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
