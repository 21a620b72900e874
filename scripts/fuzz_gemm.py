#!/usr/bin/env python3
"""Random shapes / strides / options of ops.gemm.mm and the planes GEMM (every tile selection, split K, accumulation, bias, one- and
three-product mode, transposed and column-block operands) against fp64:  python scripts/fuzz_gemm.py [n=200] [seed=0]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import padertorch_amd.ops  # noqa: F401
from padertorch_amd import _lib
from padertorch_amd.ops import gemm as G

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = 'cuda:0'
lib = _lib.load()
worst = 0.


def dim(big):
    return int(rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 64, 127, 129, 257, 300, 514, 600, 1200] + ([2400, 4800] if big else [])))


def operand(r, c, scale):
    kind = int(rng.integers(0, 4))
    if kind == 0:
        return torch.randn(r, c, device=dev) * scale
    if kind == 1:                                            # transposed storage
        return (torch.randn(c, r, device=dev) * scale).t()
    if kind == 2:                                            # column block of a wider matrix
        o = int(rng.integers(0, 5))
        return (torch.randn(r, c + 7, device=dev) * scale)[:, o:o + c]
    o = int(rng.integers(0, 5))                               # row block of a transposed wider matrix
    return (torch.randn(c + 5, r, device=dev) * scale)[o:o + c].t()


for it in range(n):
    M, N, K = dim(True), dim(True), dim(False) if rng.integers(0, 3) else int(rng.integers(1, 9000))
    x, y = operand(M, K, 2.0), operand(K, N, 0.05)
    bias = torch.randn(N, device=dev) if rng.integers(0, 2) else None
    acc = bool(rng.integers(0, 2)) and bias is None
    split = [None, 1, 2, 3, 8][int(rng.integers(0, 5))]
    tile = int(rng.integers(-1, 6))
    G.PRODUCTS = 1 if rng.integers(0, 6) == 0 else 3
    base = torch.randn(M, N + 3, device=dev)
    out = base.clone()[:, :N]
    want = x.double() @ y.double() + (bias.double() if bias is not None else 0) + (out.double() if acc else 0)
    mag = x.double().abs() @ y.double().abs() + (bias.double().abs() if bias is not None else 0) + (out.double().abs() if acc else 0)
    _lib.select_gemm_tile(tile)
    try:
        got = G.mm(x, y, bias=bias, out=out, accumulate=acc, split_k=split)
    finally:
        _lib.select_gemm_tile(-1)
    err = float(((got.double() - want).abs() / mag.clamp_min(1e-30)).max())
    tol = 6e-7 if G.PRODUCTS == 3 else 3e-3      # (hi + lo halves carry 22 bits of each operand: up to ~2 x 2^-22 per product)
    worst = max(worst, err / tol)
    ok = err < tol and torch.equal(base[:, N:], torch.cat([base[:, N:]], 1))
    print(f'{it:3d} M={M:5d} N={N:5d} K={K:5d} x{tuple(x.stride())} y{tuple(y.stride())} bias={bias is not None} acc={acc} split={split} '
          f'tile={tile} products={G.PRODUCTS} err/mag {err:.2e}{"" if ok else "   <-- FAIL"}', flush=True)
    assert ok
G.PRODUCTS = 3
print('worst err / tolerance', worst)
