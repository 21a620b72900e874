#!/usr/bin/env python3
"""End-to-end optimizer-step timings of the other BASELINE / SURVEY section 8d configurations (parity
cases, not the bench line): C3 (PIT, 64 x 4 s @ 16 kHz), C5 (deep clustering, K=3, E=20, 64 x 4 s @
16 kHz), C1-size on the GPU, and the log-mel / TasNet-loss front and back ends at C3 size.

    python scripts/bench_configs.py            ->  one JSON line per configuration
"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import padertorch_amd as pt  # noqa: E402
sys.path.insert(0, str(Path(__file__).resolve().parent))
import library_gemm_tuning as tuning  # noqa: E402  (A/B tooling outside the package)
from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel  # noqa: E402
from padertorch_amd.contrib.tcl.dc import DeepClusteringModel  # noqa: E402

dev = torch.device('cuda:0')
#: the host checks of bench.py (Trainer deferred_checks); --sync-checks for the reference's two syncs per step
DEFERRED = '--sync-checks' not in sys.argv


def timed_steps(step, trainer, warm=3, n=10):
    for _ in range(warm):
        step()
    trainer._check_pending(flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    trainer._check_pending(flush=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def pit(batch, fs, seconds, name):
    torch.manual_seed(0)
    model = PermutationInvariantTrainingModel()
    trainer = pt.Trainer(model, f'/tmp/ptmi_cfg_{name}', pt.optimizer.Adam(gradient_clipping=1.),
                         loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.), deferred_checks=DEFERRED)
    trainer.to(dev)
    trainer._flat = trainer.optimizer.use_flat_grads()
    model.train()
    n = fs * seconds
    g = torch.Generator().manual_seed(1)
    s = (0.1 * torch.randn(batch, 2, n, generator=g)).to(dev)
    y = s.sum(1)
    frames = [0]

    def step():
        feats = pt.ops.pit_features(y, s)
        frames[0] = sum(feats['num_frames'])
        loss, _, _, _ = trainer.train_step(model, feats, dev)
        loss.backward()
        trainer.optimizer_step()

    ms = timed_steps(step, trainer)
    return dict(config=name, model='PIT 3xBLSTM-600 K=2', batch=batch, fs=fs, frames_per_step=frames[0],
                ms_per_step=ms, frames_per_s=frames[0] / ms * 1e3)


def dc(batch, fs, seconds, name, K=3):
    torch.manual_seed(0)
    model = DeepClusteringModel()
    trainer = pt.Trainer(model, f'/tmp/ptmi_cfg_{name}', pt.optimizer.Adam(gradient_clipping=1.),
                         deferred_checks=DEFERRED)
    trainer.to(dev)
    trainer._flat = trainer.optimizer.use_flat_grads()
    model.train()
    n = fs * seconds
    g = torch.Generator().manual_seed(2)
    s = (0.1 * torch.randn(batch, K, n, generator=g)).to(dev)
    y = s.sum(1)
    frames = [0]

    def step():
        feats = pt.ops.pit_features(y, s)
        X = feats['X_abs'].padded                                  # [B, T, K, F]
        target = torch.nn.functional.one_hot(X.argmax(2), K).permute(0, 1, 3, 2).to(torch.float32)
        batch_ = dict(Y_abs=feats['Y_abs'], target_mask=list(target.unbind(0)), num_frames=feats['num_frames'])
        frames[0] = sum(feats['num_frames'])
        loss, _, _, _ = trainer.train_step(model, batch_, dev)
        loss.backward()
        trainer.optimizer_step()

    ms = timed_steps(step, trainer)
    return dict(config=name, model=f'DC 2xBLSTM-600 E=20 K={K}', batch=batch, fs=fs, frames_per_step=frames[0],
                ms_per_step=ms, frames_per_s=frames[0] / ms * 1e3)


if __name__ == '__main__':
    from padertorch_amd.ops import lstm as _lstm
    _lstm.DEFER_WGRAD = '--no-overlap' not in sys.argv
    if _lstm.DEFER_WGRAD:
        _lstm.warm_side_stream(dev)
    if '--default-gemms' not in sys.argv:
        tuning.use_tuned_gemms(search='--search' in sys.argv)
    for res in (pit(4, 8000, 4, 'C1 (B=4, 8 kHz)'), pit(64, 16000, 4, 'C3 (B=64, 16 kHz)'),
                dc(64, 16000, 4, 'C5 (DC, B=64, 16 kHz)')):
        print(json.dumps(res), flush=True)
