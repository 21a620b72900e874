#!/usr/bin/env python3
"""The planes GEMM (csrc/gemm_planes.hip) vs the library fp32 GEMM at the shapes of the PIT step (B = 32, T = 253: 8096 rows;
`python scripts/bench_gemm.py 32192` for the 16 kHz batch of 64): the persistent big-tile kernel with the tile the cost model
picks, every tile pinned (PTMI_GEMM_TILE / _lib.select_gemm_tile), the 128 x 128 kernel (with split K for the weight-gradient shapes), the
one-product (reduced precision) mode, the pack passes.  One JSON line per shape -> profiles/r3_gemm_microbench.jsonl."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from padertorch_amd import _lib  # noqa: E402
from padertorch_amd.ops import gemm  # noqa: E402

dev = torch.device('cuda:0')
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8096
lib = _lib.load()
PEAK = 2500. / 3          # fp32-equivalent TFLOP/s of the 16-bit dense peak at three products per product
TILES = {0: '256x320', 1: '256x256', 2: '256x192', 3: '128x320', 4: '128x256', 5: '128x128'}


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, N, K, form in [
        ('proj l0', R, 4800, 257, 'nt'), ('proj l1', R, 4800, 1200, 'nt'), ('proj l1 (hand-off k)', R, 4800, 1216, 'nt'),
        ('linear1', R, 1200, 1200, 'nt'), ('linear2', R, 514, 1200, 'nt'), ('dx lstm', R, 1200, 4800, 'nn'),
        ('dx lstm (hand-off k)', R, 1200, 4864, 'nn'), ('dx lin2', R, 1200, 514, 'nn'),
        ('dW_ih', 2400, 1200, R, 'tn'), ('dW_hh', 2400, 600, R, 'tn'), ('dW lin1', 1200, 1200, R, 'tn'),
        ('dW_ih l0', 2400, 257, R, 'tn')]:
    g = torch.Generator().manual_seed(0)
    if form == 'nt':
        x = torch.randn(M, K, generator=g).to(dev)
        w = (0.05 * torch.randn(N, K, generator=g)).to(dev)
        a, b = x, w.t()
        pa, pb = (lambda: gemm.pack_n(x)), (lambda: gemm.pack_n(w))
    elif form == 'nn':
        a = torch.randn(M, K, generator=g).to(dev)
        b = (0.05 * torch.randn(K, N, generator=g)).to(dev)
        pa, pb = (lambda: gemm.pack_n(a)), (lambda: gemm.pack_t(b))
    else:
        dg = torch.randn(K, 2 * M, generator=g).to(dev)[:, :M]
        xx = torch.randn(K, N, generator=g).to(dev)
        a, b = dg.t(), xx
        pa, pb = (lambda: gemm.pack_t(dg)), (lambda: gemm.pack_t(xx))
    out = torch.empty(M, N, device=dev)
    flop = 2.0 * M * N * K
    t_lib = timeit(lambda: torch.mm(a, b, out=out))
    rec = dict(case=name, M=M, N=N, K=K, form=form, lib_us=t_lib, lib_tflops=flop / t_lib / 1e6)
    A, Bp = pa(), pb()
    sk = gemm.auto_split_k(M, N, K)
    rec['auto_split_k'] = sk
    t = timeit(lambda: gemm.mm_planes_(out, A, Bp, M, N, K))
    rec['planes_us'], rec['planes_tflops'], rec['planes_frac'] = t, flop / t / 1e6, flop / t / 1e6 / PEAK
    ref = (a.double() @ b.double())
    rec['planes_max_err_over_mag'] = float(((out.double() - ref).abs() / (a.double().abs() @ b.double().abs())).max())
    if sk == 1:
        for tile, label in TILES.items():
            _lib.select_gemm_tile(tile)
            try:
                t = timeit(lambda: gemm.mm_planes_(out, A, Bp, M, N, K, split_k=1))
            finally:
                _lib.select_gemm_tile(-1)
            rec[f'tile_{label}_us'] = t
    else:
        for s in (1, 2, 4, 8):
            rec[f'k128_split{s}_us'] = timeit(lambda: gemm.mm_planes_(out, A, Bp, M, N, K, split_k=s))
    gemm.PRODUCTS = 1
    try:
        t = timeit(lambda: gemm.mm_planes_(out, A, Bp, M, N, K))
    finally:
        gemm.PRODUCTS = 3
    rec['one_product_us'] = t
    rec['pack_a_us'], rec['pack_b_us'] = timeit(pa), timeit(pb)
    rec['absmax_a_us'] = timeit(lambda: gemm.absmax(a))
    print(json.dumps(rec), flush=True)
