#!/usr/bin/env python3
"""Split-fp16 GEMMs (in-register split: csrc/gemm.hip; pre-split planes incl. their pack passes: csrc/gemm_planes.hip) vs
the library fp32 GEMM at the shapes of the PIT step (B = 32, T = 253: 8096 rows)."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from padertorch_amd.ops import gemm  # noqa: E402

dev = torch.device('cuda:0')
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8096


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


cases = []
for name, M, N, K, form in [
        ('proj l0', R, 4800, 257, 'nt'), ('proj l1', R, 4800, 1200, 'nt'), ('linear1', R, 1200, 1200, 'nt'),
        ('linear2', R, 514, 1200, 'nt'), ('dx lstm', R, 1200, 4800, 'nn'), ('dx lin2', R, 1200, 514, 'nn'),
        ('dW_ih', 2400, 1200, R, 'tn'), ('dW_hh', 2400, 600, R, 'tn'), ('dW lin1', 1200, 1200, R, 'tn'),
        ('dW_ih l0', 2400, 257, R, 'tn')]:
    g = torch.Generator().manual_seed(0)
    if form == 'nt':
        x = torch.randn(M, K, generator=g).to(dev)
        w = (0.05 * torch.randn(N, K, generator=g)).to(dev)
        a, b = x, w.t()
    elif form == 'nn':
        a = torch.randn(M, K, generator=g).to(dev)
        b = (0.05 * torch.randn(K, N, generator=g)).to(dev)
    else:
        dg = torch.randn(K, 2 * M, generator=g).to(dev)[:, :M]
        xx = torch.randn(K, N, generator=g).to(dev)
        a, b = dg.t(), xx
    ax, ay = gemm.absmax(a), gemm.absmax(b)
    out = torch.empty(M, N, device=dev)
    flop = 2.0 * M * N * K
    t_lib = timeit(lambda: torch.mm(a, b, out=out))
    rec = dict(case=name, M=M, N=N, K=K, form=form, lib_us=t_lib, lib_tflops=flop / t_lib / 1e6)
    for sk in ([1] if form != 'tn' else [1, 2, 4, 8]):
        t = timeit(lambda: gemm.mm(a, b, out=out, amax_x=ax, amax_y=ay, split_k=sk))
        rec[f'split_us_k{sk}'] = t
        rec[f'split_tflops_k{sk}'] = flop / t / 1e6
    rec['auto_split'] = gemm.auto_split_k(M, N, K)
    # planes GEMM: operands packed from the same sources (k-contiguous: pack_n, row-contiguous: pack_t)
    if form == 'nt':
        pa, pb = (lambda: gemm.pack_n(x, ax)), (lambda: gemm.pack_n(w, ay))
    elif form == 'nn':
        pa, pb = (lambda: gemm.pack_n(a, ax)), (lambda: gemm.pack_t(b, ay))
    else:
        pa, pb = (lambda: gemm.pack_t(dg, ax)), (lambda: gemm.pack_t(xx, ay))
    A, Bp = pa(), pb()
    t = timeit(lambda: gemm.mm_planes_(out, A, Bp, M, N, K))
    rec['planes_us'] = t
    rec['planes_tflops'] = flop / t / 1e6
    rec['pack_a_us'], rec['pack_b_us'] = timeit(pa), timeit(pb)
    ref = (a.double() @ b.double())
    rec['planes_max_err_over_mag'] = float(((out.double() - ref).abs() / (a.double().abs() @ b.double().abs())).max())
    t = timeit(lambda: gemm.mm(a, b, out=out, products=1, split_k=1))
    rec['bf16_us'] = t
    rec['absmax_us'] = timeit(lambda: gemm.absmax(a))
    print(json.dumps(rec), flush=True)
