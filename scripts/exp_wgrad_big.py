#!/usr/bin/env python3
"""Weight-gradient shapes (few output tiles, K = all rows of the batch) on the planes GEMM: the 128 x 128 slab kernel (tile 5) vs the
persistent big-tile kernel with split K (round 4: work item = (k range, tile), slabs summed by planes_reduce_kernel), every tile x
several numbers of k ranges, bf16 planes (what the backward recurrence hands on).  us per call incl. the reduction pass, fraction of
the 16-bit dense peak / 3 products, error against fp64 at the first configuration of every shape.
    python scripts/exp_wgrad_big.py [rows]  ->  profiles/r4_wgrad_big_split.txt"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from padertorch_amd import _lib  # noqa: E402

dev = torch.device('cuda:0')
lib = _lib.load()
PEAK = 2500. / 3
TILES = {0: '256x320', 1: '256x256', 2: '256x192', 3: '128x320', 4: '128x256', 5: '128x128'}
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8096


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, N in [('dW_ih', 2400, 1200), ('dW_hh', 2400, 600), ('dW_ih l0', 2400, 257), ('dW lin1', 1200, 1200), ('dW lin2', 514, 1200)]:
    g = torch.Generator().manual_seed(0)
    dg = (torch.randn(R, M, generator=g) * 0.01).to(dev)
    x = torch.randn(R, N, generator=g).to(dev)
    A = torch.ops.ptmi.pack_planes_bf16(dg, True)
    B = torch.ops.ptmi.pack_planes_bf16(x, True)
    out = torch.zeros(M, N, device=dev)
    ref = dg.double().t() @ x.double()
    mag = dg.double().abs().t() @ x.double().abs()
    flop = 2.0 * M * N * R
    print(f'## {name}: {M} x {N} x {R}')
    for tile, label in TILES.items():
        row = []
        for S in (1, 2, 3, 4, 6, 8, 12):
            if tile == 5 and S > 8:
                continue
            _lib.select_gemm_tile(tile)
            try:
                t = timeit(lambda: torch.ops.ptmi.gemm_planes_bf16_(out, A, 0, B, None, M, N, R, False, S))
                err = float(((out.double() - ref).abs() / mag).max())
            finally:
                _lib.select_gemm_tile(-1)
            row.append(f'S{S}: {t:6.1f} us {flop / t / 1e6 / PEAK:.2f}' + ('' if err < 4e-6 else f' ERR {err:.1e}'))
        print(f'{label:8s} ' + ' | '.join(row))
    t = timeit(lambda: torch.ops.ptmi.pack_planes_bf16(dg, True))
    print(f'(pack_t of the {R} x {M} operand: {t:.1f} us)')
