#!/usr/bin/env python3
"""Benchmark of the PIT mask-estimation training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torchrun)

Metric (BASELINE.json): training frames/s = mixture STFT frames through ONE full optimizer step
(on-device STFT feature front-end from HBM-resident waveforms -> BLSTM mask estimator forward ->
PIT review -> backward -> [RCCL all-reduce(sum)] -> global-norm clip -> Adam), whole job.
Workload = BASELINE.json configs[1]: 2-speaker synthetic 8 kHz mixtures, batch 32 per GPU,
4 s each (T = 253 frames/example, 8096 frames/step/GPU), PIT model defaults (3xBLSTM-600, K=2,
23 480 914 parameters), STFT 512/128.  One micro-step per rank per optimizer step (weak scaling).

The JSON line also carries
  roofline     : the kernel family with the most GPU time per step (the BLSTM recurrence), its
                 algorithmic flops / HIP-event time of its launches inside the timed steps against the
                 fp32 MFMA peak; other_kernels lists every other hand-written kernel of the step the
                 same way (the STFT front-end: 6676 B per mixture frame at K=2, SURVEY.md section 8d,
                 against the 8 TB/s HBM3E peak);
  cpu_baseline : the oracle's torch-CPU port of the reference step (oracle/torch_ref.py) timed on
                 this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FS = 8000
SECONDS = 4
BATCH = 32
K = 2
SIZE, SHIFT = 512, 128
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 dense peak (MI355X_MICROARCH.md)
FEATURE_BYTES_PER_FRAME = 3 * SHIFT * 4 + (1 + 2 * K) * (SIZE // 2 + 1) * 4   # 6676 B at K=2
PIT_LOSS_BYTES_PER_FRAME = (1 + 3 * K) * (SIZE // 2 + 1) * 4                  # 7196 B at K=2
LOSS_WEIGHTS = dict(pit_ips_loss=1., pit_mse_loss=0.)      # pit/train.py:68-71


def synthetic_batch(seed, batch, n, device):
    """K sources 0.1*N(0,1) fp32, mixture = sum (seeded, on device)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    s = 0.1 * torch.randn(batch, K, n, generator=g)
    return dict(y=s.sum(1).to(device), s=s.to(device), num_samples=[n] * batch)


def cpu_baseline(max_seconds=25.):
    """Reference algorithm (oracle/torch_ref.py: conv1d STFT, nn.LSTM on PackedSequence, python-loop
    pit_loss, clip + Adam) on the host cores, bounded sample: batch 4 x 4 s (1012 frames / step)."""
    from oracle import torch_ref
    # many-core hosts: the small LSTM GEMMs of this model run fastest on a few cores (the reference
    # README even recommends OMP_NUM_THREADS=1, pit/README.md:15); use 16 threads, state it.
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    torch.manual_seed(0)
    model = torch_ref.PITModelRef()
    opt = torch.optim.Adam(model.parameters())
    stft = torch_ref.ConvSTFT(SIZE, SHIFT)
    b = 4
    g = torch.Generator().manual_seed(1)
    s = [0.1 * torch.randn(K, FS * SECONDS, generator=g) for _ in range(b)]
    y = [x.sum(0) for x in s]

    def step():
        with torch.no_grad():
            feats = torch_ref.features_from_waveforms(stft, s, y)
        torch_ref.train_step(model, opt, [feats], LOSS_WEIGHTS, 1.)
        return sum(feats['num_frames'])

    frames = step()          # warm-up
    times = []
    t_all = time.perf_counter()
    while len(times) < 16 and (time.perf_counter() - t_all) < max_seconds:        # ~12 s of CPU work
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return dict(value=frames / med, unit='frames/s', cores=torch.get_num_threads(), kind='port',
                sample=f'batch {b} x {SECONDS} s @ {FS} Hz ({frames} frames/step), PIT defaults fp32, '
                       f'{len(times)} timed steps after 1 warm-up, median {med:.3f} s/step, '
                       f'os.cpu_count()={os.cpu_count()}')


def measured_traffic(kernel):
    """HBM bytes per launch from the committed PMC passes (profiles/r1_pmc_traffic.json: rocprofv3
    FETCH_SIZE / WRITE_SIZE in separate passes of this same command, FETCH_SIZE doubled as the
    MI355X guide prescribes for gfx950).  None when no measurement is committed."""
    f = REPO / 'profiles' / 'pmc_traffic.json'
    if not f.exists():
        return None
    return json.loads(f.read_text()).get(kernel, {}).get('hbm_bytes_per_launch')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--default-gemms', action='store_true', help='library default GEMM kernel selection')
    ap.add_argument('--sync-checks', action='store_true',
                    help='loss / grad-norm finiteness checks in the step they belong to (two host syncs per step, '
                         'the reference behaviour) instead of Trainer(deferred_checks=True)')
    ap.add_argument('--no-overlap', action='store_true',
                    help='LSTM weight gradients through autograd on the main stream (ops.lstm.DEFER_WGRAD off)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world} (launch N>1 through torchrun)'
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=device)

    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import \
        PermutationInvariantTrainingModel

    torch.manual_seed(0)
    if not args.default_gemms:
        # library GEMMs (input projections, linears, weight gradients): committed TunableOp selections
        # for these shapes (padertorch_amd/tuned/, see padertorch_amd/tuning.py); same arithmetic,
        # better hipBLASLt / rocBLAS kernels (fp32: 65-70 % -> 85-91 % of the MFMA peak)
        from padertorch_amd import tuning
        tuning.use_tuned_gemms()
    model = PermutationInvariantTrainingModel()          # defaults: F=257, 3 x BLSTM-600, K=2
    trainer = pt.Trainer(model, f'/tmp/ptmi_bench_{rank}', pt.optimizer.Adam(gradient_clipping=1.),
                         loss_weights=LOSS_WEIGHTS, virtual_minibatch_size=world,
                         deferred_checks=not args.sync_checks)
    trainer.to(device)
    trainer._flat = trainer.optimizer.use_flat_grads()
    if world > 1:
        trainer._broadcast_parameters()
    model.train()
    from padertorch_amd.ops import lstm as _lstm
    _lstm.DEFER_WGRAD = not args.no_overlap      # what Trainer.train() sets (side stream only for rocBLAS-pinned shapes)
    if _lstm.DEFER_WGRAD:
        _lstm.warm_side_stream(device)

    n = FS * SECONDS
    data = synthetic_batch(1000 + rank, BATCH, n, device)
    frames_per_step = None
    from padertorch_amd import _lib

    def step(timed):
        nonlocal frames_per_step
        # the STFT feature front-end is part of the step; inside the timed region every launch of
        # a ptmi kernel is bracketed by HIP events on the stream it runs on (torch's current stream)
        _lib.KERNEL_TIMERS = timers if timed else None
        feats = pt.ops.pit_features(data['y'], data['s'], data['num_samples'])
        frames_per_step = sum(feats['num_frames'])
        loss, _, _, _ = trainer.train_step(model, feats, device)
        loss.backward()
        trainer.optimizer_step()

    timers = []
    for _ in range(args.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    trainer._check_pending(flush=True)       # the last step's staged loss / grad-norm checks (deferred_checks)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    _lib.KERNEL_TIMERS = None
    if rank == 0:
        by_name = {}
        for name, a, b in timers:
            by_name.setdefault(name, []).append(a.elapsed_time(b))
        # One entry per hand-written kernel family seen in the timed steps: HIP-event time of every
        # launch (events recorded on the launch stream around the C-ABI call), algorithmic bytes or
        # flops per launch (SURVEY.md section 8d figures x the frames one launch processes).
        Hh, T = model.blstm.hidden_size, frames_per_step // BATCH
        rec_flop = 2.0 * 2 * BATCH * Hh * 4 * Hh * T          # both directions, one layer, one pass
        F = SIZE // 2 + 1
        spec = {
            'pit_features': ('pit_features_kernel<Plan<16,16>> (fused STFT front-end)', 'hbm',
                             FEATURE_BYTES_PER_FRAME * frames_per_step, 'pit_features'),
            'pit_pairwise_sse': ('pit_pairwise_kernel<2> (PIT mse+ips pairwise SSE)', 'hbm',
                                 PIT_LOSS_BYTES_PER_FRAME * frames_per_step, 'pit_pairwise_sse'),
            'pit_backward': ('pit_backward_kernel (d loss / d mask)', 'hbm',
                             (PIT_LOSS_BYTES_PER_FRAME + K * F * 4) * frames_per_step, 'pit_backward'),
            'lstm_forward': ('lstm_fwd_persistent_kernel (BLSTM recurrence, one launch per layer)', 'mfma',
                             rec_flop, 'lstm_fwd_persistent'),
            'lstm_backward': ('lstm_bwd_persistent_kernel (BLSTM backward-through-time, one launch per layer)',
                              'mfma', rec_flop, 'lstm_bwd_persistent'),
        }
        kernels = []
        for n, v in by_name.items():
            if n not in spec:
                continue
            label, bound, work, traffic_key = spec[n]
            ms = float(np.mean(v))
            if bound == 'hbm':
                achieved, peak, unit = work / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, 'GB/s'
            else:
                achieved, peak, unit = work / (ms * 1e-3) / 1e12, FP32_MFMA_PEAK_TFLOPS, 'TFLOP/s'
            e = dict(kernel=label, bound=bound, achieved=achieved, peak=peak, unit=unit, frac=achieved / peak,
                     traffic=measured_traffic(traffic_key), avg_launch_ms=ms,
                     launches_per_step=len(v) / args.steps, ms_per_step=float(np.sum(v)) / args.steps)
            if bound == 'hbm':
                e['algorithmic_bytes_per_launch'] = work
            else:
                e['algorithmic_flop_per_launch'] = work
                e['us_per_timestep'] = ms * 1e3 / T
            kernels.append(e)
        # the dominant kernel = the family with the most GPU time per step.  At this workload that is
        # the BLSTM recurrence: exact-fp32 matrix-core work against the 157.3 TFLOP/s fp32 MFMA peak,
        # bound by the per-timestep dependency chain, not by the matrix cores.
        kernels.sort(key=lambda e: -e['ms_per_step'])
        dominant, other = kernels[0], kernels[1:]
        out = {
            'metric': 'training frames/sec (PIT mask-est, 2-spk 8 kHz)',
            'value': frames_per_step * world * args.steps / elapsed,
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {
                'workload': f'BASELINE configs[1]: PIT mask estimator (3xBLSTM-600, K=2, 23.5M params), '
                            f'{BATCH} x {SECONDS} s 2-spk {FS} Hz mixtures per GPU '
                            f'({frames_per_step} frames/step/GPU), STFT {SIZE}/{SHIFT} on device, '
                            f'full optimizer step (Adam, clip 1)',
                'global_batch': BATCH * world,
                'frames_per_step': frames_per_step * world,
                'parallelism': f'dp{world}',
                'blstm': 'HIP recurrence (csrc/lstm.hip)' if model.hip_blstm else 'torch.nn.LSTM (MIOpen)',
                'gemms': 'library defaults' if args.default_gemms else 'hipBLASLt/rocBLAS fp32, TunableOp selections (padertorch_amd/tuned)',
                'host_checks': 'same step (2 syncs)' if args.sync_checks else 'loss / grad-norm finiteness inspected one step late, optimizer update gated on the device (Trainer deferred_checks)',
                'lstm_weight_gradients': 'autograd, main stream' if args.no_overlap else 'in place, side stream next to the next recurrence (rocBLAS-pinned shapes)',
            },
            'roofline': dominant,
            'other_kernels': other,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
