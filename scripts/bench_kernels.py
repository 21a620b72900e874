#!/usr/bin/env python3
"""Per-kernel microbenchmarks (HIP-event timed, many launches back to back) with roofline figures.

    python scripts/bench_kernels.py [--big]

Algorithmic bytes per frame follow SURVEY.md section 8d: STFT 2568 B, iSTFT 2568 B, fused PIT
front-end 6676 B, fused PIT loss 7196 B (K=2, F=257, fp32 in / complex64 out).
"""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import padertorch_amd as pt  # noqa: E402
from padertorch_amd.ops.losses import pit_mse_ips_losses  # noqa: E402

PEAK = 8000.0


def timeit(fn, iters=30, warm=3):
    """Mean GPU time of the ptmi launches inside fn (events bracket each C-ABI call), in us."""
    from padertorch_amd import _lib
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    _lib.KERNEL_TIMERS = []
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    t = sum(a.elapsed_time(b) for _, a, b in _lib.KERNEL_TIMERS) / iters * 1e3
    _lib.KERNEL_TIMERS = None
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--big', action='store_true')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    res = []
    cfgs = [(32, 32000), (64, 64000)] + ([(512, 64000)] if args.big else [])
    for B, N in cfgs:
        K = 2
        g = torch.Generator().manual_seed(0)
        s = (0.1 * torch.randn(B, K, N, generator=g)).to(dev)
        y = s.sum(1)
        st = pt.ops.STFT(512, 128)
        T = st._frames_for(N)
        x3 = torch.cat([y[:, None], s], 1).reshape(-1, N).contiguous()     # (K+1)*B rows
        X = st(x3)
        rows = x3.shape[0]
        t = timeit(lambda: st(x3))
        res.append(dict(kernel='stft_fwd', B=rows, N=N, frames=rows * T, us=t, GBs=rows * T * 2568 / t / 1e3))
        t = timeit(lambda: st.inverse(X))
        res.append(dict(kernel='istft', B=rows, N=N, frames=rows * T, us=t, GBs=rows * T * 2568 / t / 1e3))
        t = timeit(lambda: pt.ops.pit_features(y, s))
        res.append(dict(kernel='pit_features (+ packed log1p fp32 and fp16 planes: 8856 B / frame)', B=B, N=N, frames=B * T, us=t,
                        GBs=B * T * 8856 / t / 1e3))
        t = timeit(lambda: pt.ops.pit_features(y, s, packed_log1p=False))
        res.append(dict(kernel='pit_features (plain: 6676 B / frame)', B=B, N=N, frames=B * T, us=t, GBs=B * T * 6676 / t / 1e3))
        f = pt.ops.pit_features(y, s)
        mask = torch.rand(B, T, K, 257, device=dev, requires_grad=True)
        Y, Xa, C = f['Y_abs'].padded, f['X_abs'].padded, f['cos_phase_difference'].padded
        t = timeit(lambda: pit_mse_ips_losses(mask, Y, Xa, C))
        res.append(dict(kernel='pit_loss_fwd(3 kernels)', B=B, frames=B * T, us=t, GBs=B * T * 7196 / t / 1e3))
        loss = pit_mse_ips_losses(mask, Y, Xa, C)[0]
        t = timeit(lambda: torch.autograd.grad(loss[1], mask, retain_graph=True))
        res.append(dict(kernel='pit_loss_bwd', B=B, frames=B * T, us=t,
                        GBs=B * T * (7196 + 2056) / t / 1e3))
        # time-domain PIT losses (TasNet trio) from one statistics pass: 2*K*N*4 bytes read per example;
        # backward: read 2*K*N*4, write K*N*4
        from padertorch_amd.ops.losses import regression
        x = (s.flip(1) + 0.01 * torch.randn_like(s)).requires_grad_(True)
        t = timeit(lambda: regression.pair_stats(x.detach(), s))
        res.append(dict(kernel='td_pair_stats', B=B, N=N, us=t, GBs=B * 2 * K * N * 4 / t / 1e3))
        tot = sum(v[0].sum() for v in regression.pit_td_losses(x, s).values())
        t = timeit(lambda: torch.autograd.grad(tot, x, retain_graph=True))
        res.append(dict(kernel='td_lincomb (backward)', B=B, N=N, us=t, GBs=B * 3 * K * N * 4 / t / 1e3))
    # unit-norm embeddings (dc.py:70) at the C5 batch: 8 E F bytes per row forward, 12 E F backward
    for rows in ((8096, 32192) if args.big else (8096,)):
        E, F = 20, 257
        x = torch.randn(rows, E, F, device=dev, requires_grad=True)
        gy = torch.randn(rows, E, F, device=dev)
        t = timeit(lambda: pt.ops.unit_norm(x.detach()))
        res.append(dict(kernel='unit_norm_fwd', rows=rows, us=t, GBs=rows * E * F * 8 / t / 1e3))
        yy = pt.ops.unit_norm(x)
        t = timeit(lambda: torch.autograd.grad(yy, x, gy, retain_graph=True))
        res.append(dict(kernel='unit_norm_bwd', rows=rows, us=t, GBs=rows * E * F * 12 / t / 1e3))
    for r in res:
        r['frac_of_8TBs'] = r['GBs'] / PEAK
        print(json.dumps(r))


if __name__ == '__main__':
    main()
