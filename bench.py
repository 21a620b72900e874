#!/usr/bin/env python3
"""Benchmark of the PIT mask-estimation training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5] [--dry]

``--gpus N`` with N > 1 launches itself as N ranks (``python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1``, one rank per GPU over RCCL); when it is already running under
torchrun (RANK / WORLD_SIZE set) it is one of those ranks.

Metric (BASELINE.json): training frames/s = mixture STFT frames through ONE full optimizer step
(on-device STFT feature front-end from HBM-resident waveforms -> BLSTM mask estimator forward ->
PIT review -> backward -> [RCCL all-reduce(sum), per layer bucket under the backward pass] -> global-norm
clip -> Adam), whole job.  Default workload = BASELINE.json configs[1]: 2-speaker synthetic 8 kHz mixtures,
batch 32 per GPU, 4 s each (T = 253 frames/example, 8096 frames/step/GPU), PIT model defaults (3xBLSTM-600,
K=2, 23 480 914 parameters), STFT 512/128; one micro-step per rank per optimizer step (weak scaling).
Other configurations of BASELINE.json (parity / scaling cases, not the default line):
  c3  PIT, batch 64, 16 kHz;  c4  c3 with 4 micro-steps per rank per optimizer step (the reference's
  virtual_minibatch_size = 4 x devices);  c5  deep-clustering model, K = 3, batch 64, 16 kHz.

The JSON line also carries
  roofline        the ONE kernel with the most GPU time per step (as rocprofv3's summary of the same command names it), HIP-event
                  time of its launches inside the timed steps minus the bracket of an empty launch, against the peak that bounds
                  it; other_kernels lists every other hand-written kernel of the step the same way, plus the stand-alone STFT /
                  iSTFT op at a chip-filling batch; roofline_family = all planes GEMM launches of the step together;
  value_ragged    the same step on SURVEY 8d's training distribution (example lengths ~ U[3 s, 6 s]) in frames/s, ms_per_step_ragged;
  cpu_baseline    the oracle's torch-CPU port of the reference step (oracle/torch_ref.py) timed on this box's host
                  cores on a bounded sample (rank 0, N = 1 only), at 1 thread, 16 threads and all cores;
  value / ms_per_step   N = 1: the CAPTURED step (train.graphed.GraphedStep: every launch of the optimizer step in one hipGraph, replayed)
                  with the loss / gradient-norm / watchdog checks at the end of the SAME step - one host synchronisation per step, errors
                  raise in the iteration they belong to like the reference's (trainer.py:622-636, :740-780);
  ms_per_step_eager_deferred / _end_of_step_checks / _sync_checks   the eager step of rounds 1-4 with the checks one step late (the host
                  runs ahead), at the end of the same step, and with the reference's two mid-step syncs;
  ms_per_step_h2d the same step with the waveform batch starting in (pinned) host memory;
  rccl            (N > 1) world size, bucket layout and the time of a blocking all-reduce of the flat gradient buffer.
``--dry`` (no GPU needed; used by the CPU test suite): the same launcher, process group (gloo), gradient buckets
and JSON line around a stub step.
"""
import argparse
import json
import os
import socket
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

SIZE, SHIFT = 512, 128
SECONDS = 4
TIMER_EVERY = 8                # kernel launches carry HIP events in every 8th timed step (see step())
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 dense peak (MI355X_MICROARCH.md)
FP16_MFMA_PEAK_TFLOPS = 2500.  # v_mfma_f32_32x32x16_f16 dense peak (MI355X_MICROARCH.md)
LOSS_WEIGHTS = dict(pit_ips_loss=1., pit_mse_loss=0.)      # pit/train.py:68-71

CONFIGS = {
    'c2': dict(model='pit', batch=32, fs=8000, K=2, micro=1,
               label='BASELINE configs[1]: PIT mask estimator (3xBLSTM-600, K=2, 23.5M params)'),
    'c3': dict(model='pit', batch=64, fs=16000, K=2, micro=1,
               label='BASELINE configs[2]: PIT mask estimator, batch 64, 16 kHz'),
    'c4': dict(model='pit', batch=64, fs=16000, K=2, micro=4,
               label='BASELINE configs[3]: PIT mask estimator, batch 64 per GPU, 16 kHz, 4 micro-steps per rank per '
                     'optimizer step (virtual_minibatch_size = 4 x GPUs)'),
    'c5': dict(model='dc', batch=64, fs=16000, K=3, micro=1,
               label='BASELINE configs[4]: deep-clustering model (2xBLSTM-600, E=20), K=3, batch 64, 16 kHz'),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='c2')
    ap.add_argument('--dry', action='store_true', help='launcher / process group / buckets / JSON line around a stub step (CPU, gloo)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-variant', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-baseline-batch', type=int, default=4, help=argparse.SUPPRESS)
    ap.add_argument('--no-kernel-events', action='store_true',
                    help='no eager pass with per-kernel HIP events behind the timed replays (profiling runs: the trace then holds the '
                         'warm-up steps and the replays only)')
    ap.add_argument('--no-extras', action='store_true', help='skip the sync-checks / host-to-device / all-reduce side measurements')
    ap.add_argument('--library-gemms', action='store_true',
                    help='dense layers on the BLAS library (TunableOp selections of scripts/tuned) instead of csrc/gemm.hip')
    ap.add_argument('--bf16', action='store_true',
                    help="BASELINE configs[1]'s reduced-precision run: plain 16-bit operands in the dense layers (the hi halves only: fp16 "
                         'for activations and weights, bf16 for the gate gradients), fp32 accumulation; not the default')
    ap.add_argument('--sync-checks', action='store_true',
                    help='loss / grad-norm finiteness checks in the step they belong to (two host syncs per step, '
                         'the reference behaviour) instead of Trainer(deferred_checks=True)')
    ap.add_argument('--eager', action='store_true',
                    help='time the eager step with deferred checks (rounds 1-4) instead of the captured step (train.graphed.GraphedStep: one '
                         'hipGraph per optimizer step, loss / grad-norm checks at the end of the SAME step); N > 1, --ragged and --sync-checks '
                         'are eager anyway')
    ap.add_argument('--no-overlap', action='store_true',
                    help='LSTM weight gradients through autograd on the main stream (ops.lstm.DEFER_WGRAD off)')
    ap.add_argument('--ragged', action='store_true',
                    help="SURVEY 8d's training distribution: example lengths ~ U[3 s, 6 s] (sorted, zero-padded waveforms) instead of "
                         'the fixed 4 s of the headline; a reported mode (no roofline entry)')
    ap.add_argument('--no-overlap-allreduce', action='store_true', help='one all-reduce of the flat buffer in optimizer_step')
    ap.add_argument('--dp-graph', action='store_true',
                    help='N = 1: run under a process group of ONE rank (RCCL) and time the captured DATA-PARALLEL step (two graphs with the '
                         "'flat+words' exchange between them: train.graphed split_for_allreduce) - what N > 1 probes as schedule graph_split")
    ap.add_argument('--no-graph-split', action='store_true', help='N > 1: do not probe the captured data-parallel step')
    ap.add_argument('--capture-rccl', action='store_true',
                    help="with --dp-graph: ONE graph per step that contains the layer buckets' RCCL all-reduces (Trainer.graph_exchange = "
                         "'captured': opt-in, validated with a one-rank group only; not probed at N > 1 - a hang there would cost the run)")
    ap.add_argument('--row-slots', action='store_true',
                    help='with --ragged: 2 x batch examples end to end in `batch` row slots (model.row_slots), the timed step itself '
                         '(the default line reports the same as value_ragged_row_slots)')
    return ap.parse_args(argv)


def self_launch(args):
    """``python bench.py --gpus N`` -> N ranks under torch.distributed.run (replaces this process)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def synthetic_batch(seed, batch, K, n, device, lengths=None):
    """K sources 0.1*N(0,1) fp32, mixture = sum (seeded); ``lengths``: samples per example (descending; rows
    zero past their length), default all ``n``."""
    import torch
    g = torch.Generator(device='cpu').manual_seed(seed)
    s = 0.1 * torch.randn(batch, K, n, generator=g)
    if lengths is not None:
        for b, nb in enumerate(lengths):
            s[b, :, nb:] = 0.
    return dict(y=s.sum(1).to(device), s=s.to(device), num_samples=list(lengths) if lengths is not None else [n] * batch)


def cpu_baseline_variant(threads, max_seconds=12., b=4):
    """One thread count of the CPU baseline (runs in its own process: see cpu_baseline)."""
    import numpy as np
    import torch
    from oracle import torch_ref
    fs, K = 8000, 2
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = torch_ref.PITModelRef()
    opt = torch.optim.Adam(model.parameters())
    stft = torch_ref.ConvSTFT(SIZE, SHIFT)
    g = torch.Generator().manual_seed(1)
    s = [0.1 * torch.randn(K, fs * SECONDS, generator=g) for _ in range(b)]
    y = [x.sum(0) for x in s]

    def step():
        with torch.no_grad():
            feats = torch_ref.features_from_waveforms(stft, s, y)
        torch_ref.train_step(model, opt, [feats], LOSS_WEIGHTS, 1.)
        return sum(feats['num_frames'])

    times = []
    if b > 4:                # (the GPU's batch: a step takes ~1 min on this host whatever the thread count - no warm-up step, up to 5 timed)
        max_seconds = 50.
        t0 = time.perf_counter()
        frames = step()
        times.append(time.perf_counter() - t0)
    else:
        frames = step()      # warm-up
    t_all = time.perf_counter() - sum(times)
    while len(times) < (12 if b <= 4 else 5) and (time.perf_counter() - t_all) < max_seconds:
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return dict(cores=threads, batch=b, value=frames / med, steps=len(times), s_per_step=med, frames_per_step=frames)


def cpu_baseline(max_seconds=12., variant_timeout=60.):
    """Reference algorithm (oracle/torch_ref.py: conv1d STFT, nn.LSTM on PackedSequence, python-loop pit_loss, clip +
    Adam) on the host cores, bounded sample: batch 4 x 4 s at 8 kHz (1012 frames / step = BASELINE configs[0]) at three thread
    counts: 1 (the reference's README recommends OMP_NUM_THREADS=1, pit/README.md:15), 16 and min(64, cores) (the small LSTM
    GEMMs of this model thrash when spread over hundreds of threads: round 2's all-cores variant never finished its first step),
    plus the GPU's own batch of 32 once at the best of those thread counts.  Every variant runs in its own process with a hard
    time limit and is reported as not finished instead of stalling the benchmark.  ``value`` = the best batch-4 figure."""
    import subprocess
    ncpu = os.cpu_count() or 1

    def run(threads, batch, limit):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
            env.pop(k, None)
        try:
            p = subprocess.run([sys.executable, str(Path(__file__).resolve()), '--cpu-baseline-variant', str(threads),
                                '--cpu-baseline-batch', str(batch)], capture_output=True, text=True, timeout=limit, env=env)
            line = [l for l in p.stdout.splitlines() if l.startswith('{')]
            return json.loads(line[-1]) if line else dict(cores=threads, batch=batch, value=None, note=f'failed: {p.stderr[-200:]}')
        except subprocess.TimeoutExpired:
            return dict(cores=threads, batch=batch, value=None, note=f'warm-up + one step did not finish within {limit:.0f} s')

    variants = [run(threads, 4, variant_timeout) for threads in sorted({1, min(16, ncpu), min(64, ncpu)})]
    done = [v for v in variants if v.get('value')]
    best = max(done, key=lambda v: v['value'])
    # the GPU run's own batch (32 x 4 s: a step is ~8 x the batch-4 one) at OMP_NUM_THREADS=1 (pit/README.md:15) and at 8 / 16 / 32
    # threads - "the CPU at the GPU's batch" as a number with its best thread count.  The four legs run SIDE BY
    # SIDE, each in its own process on its own threads (57 of the host's cores), so that the bench stays within minutes; each takes up
    # to 5 timed steps within its limit and reports how many it got.
    from concurrent.futures import ThreadPoolExecutor
    legs = sorted({1, min(8, ncpu), min(16, ncpu), min(32, ncpu)})
    if sum(legs) <= ncpu:
        with ThreadPoolExecutor(len(legs)) as pool:
            b32 = list(pool.map(lambda t: run(t, 32, 2 * variant_timeout), legs))
    else:
        b32 = [run(t, 32, 2 * variant_timeout) for t in legs[-1:]]
    done32 = [v for v in b32 if v.get('value')]
    best32 = max(done32, key=lambda v: v['value']) if done32 else None
    return dict(value=best['value'], unit='frames/s', cores=best['cores'], kind='port',
                sample=f'batch 4 x {SECONDS} s @ 8000 Hz ({best["frames_per_step"]} frames/step), PIT defaults fp32, median of up to 12 '
                       f'timed steps (<= {max_seconds:.0f} s) after 1 warm-up per thread count (own process each), os.cpu_count()={ncpu}',
                variants=variants, batch32=best32, batch32_variants=b32,
                batch32_sample=f'batch 32 x {SECONDS} s @ 8000 Hz (8096 frames/step = the GPU run\'s step), up to 5 timed steps (<= 50 s, no warm-up step: one step takes about a minute) '
                               f'per thread count, thread counts {legs} side by side')


def replay_profile(symbol, config):
    """Average launch duration (ms) of the kernel whose name contains ``symbol`` in the committed rocprofv3 summary of THIS command's
    replays (``profiles/r6_kernel_trace_replay_<config>.txt``: ``rocprofv3 --kernel-trace --stats -- python bench.py --config <config>
    --no-extras --no-kernel-events --no-cpu-baseline``: the warm-up steps and the replayed graph, no eager measurement passes).  A
    replay takes no event records between its nodes (external events are refused on ROCm: ``scripts/mb/graph_events.py``), so this
    is where the kernels of the TIMED mode are timed; None without a committed profile."""
    f = REPO / 'profiles' / f'r6_kernel_trace_replay_{config}.txt'
    if not f.exists():
        return None
    for line in f.read_text().splitlines():
        parts = line.split(None, 11)
        if len(parts) == 12 and symbol in parts[11] and parts[0].isdigit():
            return float(parts[2]) * 1e-3
    return None


def measured_traffic(kernel):
    """HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json: rocprofv3 FETCH_SIZE /
    WRITE_SIZE in separate passes of this same command, FETCH_SIZE doubled as the MI355X guide prescribes for
    gfx950).  None when no measurement is committed."""
    f = REPO / 'profiles' / 'pmc_traffic.json'
    if not f.exists():
        return None
    return json.loads(f.read_text()).get(kernel, {}).get('hbm_bytes_per_launch')


SYMBOLS = {'pit_features': 'pit_features_kernel', 'pit_pairwise_sse': 'pit_pairwise_kernel', 'pit_backward': 'pit_backward_kernel',
           'lstm_forward': 'lstm_fwd_daf_kernel', 'lstm_backward': 'lstm_bwd_split_kernel'}
TILE_NAMES = {0: '8, 5 (256 x 320)', 1: '8, 4 (256 x 256)', 2: '8, 3 (256 x 192)', 3: '4, 5 (128 x 320)', 4: '4, 4 (128 x 256)'}


def event_bracket_overhead_ms(device, n=200):
    """What two HIP events around ONE launch measure when the kernel does nothing: the launch's own dispatch latency sits
    inside every event bracket, rocprofv3's kernel durations do not contain it.  A 30 us HBM kernel priced on the raw
    bracket looks ~20 % slower than in the profile of the same command; `kernel_report` subtracts this figure so that the
    two tools quote one number (the raw bracket stays in the entry as ``avg_launch_ms_events``)."""
    import torch
    from padertorch_amd import _lib
    hooks = _lib.test_hooks()          # (the empty launch lives in the test-hook library, not in libptmi.so)
    st = _lib.stream(device)
    for _ in range(20):
        hooks.ptmi_test_occupy(1, 64, 0, 0, st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        hooks.ptmi_test_occupy(1, 64, 0, 0, st)
        b.record()
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ev)
    return v[len(v) // 2]


def standalone_front_end(device, overhead_ms, rows=192, samples=64000):
    """north_star prices 'the STFT kernel' against the HBM peak: the forward and inverse STFT op alone on a batch that fills the
    chip (192 x 8 s @ 8 kHz: 97 k frames, 2568 algorithmic bytes per frame each way), HIP events per launch minus the empty-launch
    bracket.  Not part of the training step (its fused front-end is pit_features): reported as stand-alone rows."""
    import torch
    import padertorch_amd as pt
    stft = pt.ops.STFT(SIZE, SHIFT, complex_representation='stacked')
    x = torch.randn(rows, samples, device=device)
    out = []
    with torch.no_grad():
        spec = stft(x)
        frames = spec.shape[1] * rows
        for name, label, fn in (('stft_fwd', 'stft_fwd_kernel<Plan<16,16>> (ptmi_stft_forward, stand-alone)', lambda: stft(x)),
                                ('istft', 'istft_kernel<Plan<16,16>> (ptmi_istft_forward, stand-alone)', lambda: stft.inverse(spec))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ms = []
            for _ in range(10):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            raw = sorted(ms)[len(ms) // 2]
            t = max(raw - overhead_ms, 1e-6)
            work = (SHIFT * 4 + (SIZE // 2 + 1) * 8) * frames
            resident = work <= 256 * 2 ** 20
            out.append(dict(kernel=label, bound='hbm', achieved=work / (t * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s',
                            frac=work / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=None, avg_launch_ms=t, avg_launch_ms_events=raw,
                            standalone=True, frames_per_launch=frames, algorithmic_bytes_per_launch=work,
                            workload=f'{rows} x {samples} samples, STFT {SIZE}/{SHIFT}',
                            memory=('fits the 256 MiB Infinity Cache (input freshly written): a cache-resident figure, NOT an HBM one' if resident
                                    else 'far beyond the 256 MiB Infinity Cache: an HBM figure')))
    return out


def kernel_report(timers, steps, cfg, frames_per_step, hidden, micro, products_mode=3, overhead_ms=0., config=None):
    """One entry per hand-written KERNEL seen in the timed steps, named as rocprofv3's summary of the same command names it (the
    planes GEMMs by kernel, plane type and workgroup tile: ``ptmi_gemm_planes_plan`` tells which one a call runs as): HIP-event time
    of every launch (events recorded on the launch stream around the C-ABI call) minus the bracket of an empty launch, against the
    algorithmic bytes or flops of the launch.  Sorted by GPU time per step: the first entry is the step's top kernel."""
    import numpy as np
    from padertorch_amd import _lib
    lib = _lib.load()
    by_name = {}
    for name, a, b in timers:
        by_name.setdefault(name, []).append(a.elapsed_time(b))
    K, B = cfg['K'], cfg['batch']
    F = SIZE // 2 + 1
    T = frames_per_step // B
    fpl = frames_per_step            # frames one launch processes (one micro-step)
    rec_flop = 2.0 * 2 * B * hidden * 4 * hidden * T          # both directions, one layer, one pass
    # per frame: (1 + K) x shift samples in, (1 + 2 K) x F magnitudes / cosines out, and the first layer's input the kernel writes
    # itself: F fp32 log-magnitudes in PackedSequence order + their fp16 (hi, lo) planes (F rounded up to 32-wide blocks)
    feature_bytes = ((1 + K) * SHIFT * 4 + (1 + 2 * K) * F * 4 + F * 4 + (F + 31) // 32 * 32 * 2 * 2) * fpl
    pit_bytes = (1 + 3 * K) * F * 4 * fpl
    rec_note = ('fp16/bf16 MFMA dense peak 2500 TFLOP/s / 3 products.  What bounds the kernel is the serial per-timestep hand-off (stores '
                'becoming visible, then the gather of the chain\'s rows: 39 KB forward, 154 KB backward per CU and step = the CU\'s '
                'vector-memory issue rate, independent of the batch), not the matrix cores (DESIGN.md section 3.3)')
    spec = {
        'pit_features': ('pit_features_kernel<Plan<16,16>> (fused STFT front-end incl. the packed log-magnitude input of the first layer)', 'hbm', feature_bytes),
        'pit_pairwise_sse': ('pit_pairwise_kernel (PIT mse+ips pairwise SSE)', 'hbm', pit_bytes),
        'pit_backward': ('pit_backward_kernel (d loss / d mask)', 'hbm', pit_bytes + K * F * 4 * fpl),
        'lstm_forward': ('lstm_fwd_daf_kernel (BLSTM recurrence, one persistent launch per layer, fp16 hi/lo MFMA products, data-as-flag '
                         'hand-off)', 'mfma16x3', rec_flop),
        'lstm_backward': ('lstm_bwd_split_kernel<DAF> (BLSTM backward-through-time, one persistent launch per layer, bf16 hi/lo MFMA '
                          'products, data-as-flag hand-off; hands dgates^T on as bf16 planes)', 'mfma16x3', rec_flop),
    }
    kernels = []
    gemm, packs = {}, {}
    family = dict(flop=0., ms=0., launches=0)
    for n, v in by_name.items():
        if n.startswith(('gemm_planes:', 'gemm_planes_bf16:')):          # gemm_planes[_bf16]:MxNxK:split
            parts = n.split(':')
            M, N, Kd = (int(x) for x in parts[1].split('x'))
            bf16 = n.startswith('gemm_planes_bf16')
            split = int(parts[2]) if len(parts) > 2 and parts[2].lstrip('-').isdigit() else 1
            plan = int(lib.ptmi_gemm_planes_plan(M, N, Kd, split))
            tile, ranges = plan // 100, plan % 100
            dt = 'bf16' if bf16 else 'fp16'
            one = ', one product' if products_mode == 1 else ''
            if tile == 5:
                key = (f'gemm_planes_kernel<{dt}{one}> (128 x 128, slab split K' + (', co-resident with a recurrence' if split < 0 else '') + ')',
                       products_mode, 'weight gradients beside a backward recurrence')
            else:
                key = (f'gemm_planes_big_kernel<{dt}, {TILE_NAMES[tile]}{one}>' + (' split K + planes_reduce_kernel' if ranges > 1 else ''),
                       products_mode, 'persistent big-tile kernel')
            e = gemm.setdefault(key, dict(flop=0., ms=0., launches=0, shapes=set()))
            e['flop'] += 2.0 * M * N * Kd * len(v)
            e['ms'] += float(np.sum(v)) - overhead_ms * len(v)
            e['launches'] += len(v)
            e['shapes'].add(f'{M}x{N}x{Kd}')
            family['flop'] += 2.0 * M * N * Kd * len(v)
            family['ms'] += float(np.sum(v)) - overhead_ms * len(v)
            family['launches'] += len(v)
            continue
        if n.startswith('pack_planes'):          # pack_planes_t:KxC / pack_planes_n:RxK: 4 B read + 4 B written per element
            a_, b_ = (int(x) for x in n.split(':')[1].split('x'))
            e = packs.setdefault('pack', dict(bytes=0., ms=0., launches=0))
            e['bytes'] += 8.0 * a_ * b_ * len(v)
            e['ms'] += float(np.sum(v)) - overhead_ms * len(v)
            e['launches'] += len(v)
            continue
        if n not in spec:
            continue
        label, bound, work = spec[n]
        raw = float(np.mean(v))
        ms = max(raw - overhead_ms, 1e-6)
        if bound == 'hbm':
            achieved, peak, unit = work / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, 'GB/s'
        else:
            # the recurrence multiplies with 16-bit MFMA, three products per fp32 product: algorithmic flop against the
            # 16-bit dense peak / 3 (its real bound is the per-timestep hand-off chain)
            achieved, peak, unit = work / (ms * 1e-3) / 1e12, FP16_MFMA_PEAK_TFLOPS / 3, 'TFLOP/s'
        e = dict(kernel=label, bound='hbm' if bound == 'hbm' else 'mfma', achieved=achieved, peak=peak, unit=unit,
                 frac=achieved / peak, traffic=measured_traffic(n), avg_launch_ms=ms, avg_launch_ms_events=raw,
                 launches_per_step=len(v) / steps, ms_per_step=ms * len(v) / steps)
        if bound == 'hbm':
            e['algorithmic_bytes_per_launch'] = work
        else:
            e['algorithmic_flop_per_launch'] = work
            e['us_per_timestep'] = ms * 1e3 / T
            e['frac_of_fp32_mfma_peak'] = achieved / FP32_MFMA_PEAK_TFLOPS       # the measure of round 1 (exact-fp32 MFMA kernels)
            e['peak_note'] = rec_note
        # the same kernel as rocprofv3 times it INSIDE the replayed graph (committed profile of this command; the event bracket above is
        # taken in an eager pass beside other queues' work): what `frac` becomes on that duration
        rp = replay_profile(SYMBOLS[n], config) if config else None
        if rp:
            e['avg_launch_ms_replay_profile'] = rp
            e['frac_replay_profile'] = work / (rp * 1e-3) / (1e9 if bound == 'hbm' else 1e12) / peak
            e['replay_profile'] = f'profiles/r6_kernel_trace_replay_{config}.txt'
        kernels.append(e)
    for (label, products, what), e in gemm.items():
        if not e['launches']:
            continue
        achieved = e['flop'] / (e['ms'] * 1e-3) / 1e12
        peak = FP16_MFMA_PEAK_TFLOPS / products
        kernels.append(dict(
            kernel=f'{label}: {what}', bound='mfma', achieved=achieved, peak=peak, unit='TFLOP/s', frac=achieved / peak,
            traffic=measured_traffic('gemm_planes'),
            peak_note=f'fp16/bf16 MFMA dense peak {FP16_MFMA_PEAK_TFLOPS:.0f} TFLOP/s / {products} products; achieved = algorithmic 2MNK flop '
                      f'of the launches / their HIP-event time inside the step (weight gradients run beside a recurrence)',
            avg_launch_ms=e['ms'] / e['launches'], launches_per_step=e['launches'] / steps, ms_per_step=e['ms'] / steps,
            algorithmic_flop_per_step=e['flop'] / steps, shapes=sorted(e['shapes'])))
    for e in packs.values():
        achieved = e['bytes'] / (e['ms'] * 1e-3) / 1e9
        kernels.append(dict(
            kernel='pack_planes_t_kernel / pack_planes_n_kernel (fp32 operand -> 16-bit (hi, lo) planes in MFMA fragment order)',
            bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s', frac=achieved / HBM_PEAK_GBS, traffic=None,
            avg_launch_ms=e['ms'] / e['launches'], launches_per_step=e['launches'] / steps, ms_per_step=e['ms'] / steps,
            algorithmic_bytes_per_step=e['bytes'] / steps))
    kernels.sort(key=lambda e: -e['ms_per_step'])
    fam = None
    if family['launches']:
        achieved = family['flop'] / (family['ms'] * 1e-3) / 1e12
        peak = FP16_MFMA_PEAK_TFLOPS / products_mode
        fam = dict(kernel='all planes GEMM launches of the step (gemm_planes_big_kernel + gemm_planes_kernel, both queues)', bound='mfma',
                   achieved=achieved, peak=peak, unit='TFLOP/s', frac=achieved / peak, launches_per_step=family['launches'] / steps,
                   ms_per_step=family['ms'] / steps, algorithmic_flop_per_step=family['flop'] / steps)
    return kernels, fam


def main():
    args = parse_args()
    if args.dp_graph:
        assert args.gpus == 1, '--dp-graph: the one-rank process group of a single-GPU run (N > 1 probes the same step as schedule graph_split)'
        args.no_extras = True
    assert not args.capture_rccl or args.dp_graph, '--capture-rccl goes with --dp-graph (N = 1)'
    if os.environ.get('PTMI_BENCH_TRACE'):
        import faulthandler
        faulthandler.dump_traceback_later(40, repeat=True, file=sys.stderr)
    if args.cpu_baseline_variant:
        print(json.dumps(cpu_baseline_variant(args.cpu_baseline_variant, b=args.cpu_baseline_batch)), flush=True)
        return
    # stdout carries ONE JSON line: RCCL's log (the image exports NCCL_DEBUG=VERSION: a five-line banner) goes to stderr
    os.environ.setdefault('NCCL_DEBUG_FILE', '/dev/stderr')
    if os.environ.get('NCCL_DEBUG') == 'VERSION':        # (printed with printf at process exit, behind the JSON line, whatever NCCL_DEBUG_FILE says)
        del os.environ['NCCL_DEBUG']
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    cfg = CONFIGS[args.config]
    if args.dry:
        device = torch.device('cpu')
        backend = 'gloo'
    else:
        assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs (--dry for the launcher check on CPU)'
        # PTMI_BENCH_SHARE_GPU=1 (a functional pre-flight of the N > 1 code path on a ONE-GPU box: every rank on cuda:0, the collectives
        # through gloo - the timings mean nothing, the ranks' recurrences take turns on the CUs): not what the driver runs
        share = bool(os.environ.get('PTMI_BENCH_SHARE_GPU'))
        dev_index = 0 if share else local_rank
        torch.cuda.set_device(dev_index)
        device = torch.device('cuda', dev_index)
        backend = 'gloo' if share else 'nccl'
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.dry or backend == 'gloo':
            dist.init_process_group(backend)
        else:
            dist.init_process_group(backend, device_id=device)
    elif args.dp_graph and not args.dry:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        dist.init_process_group(backend, init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=device)

    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    from padertorch_amd.contrib.tcl.dc import DeepClusteringModel
    from padertorch_amd.ops import gemm as _gemm
    from padertorch_amd.ops import lstm as _lstm

    torch.manual_seed(0)
    # the host side of the step is a few small CPU tensor operations; with the default (one intra-op thread per core: 256 here) each
    # of them that exceeds torch's grain size wakes the whole pool - milliseconds on a box that other tenants load
    torch.set_num_threads(min(8, os.cpu_count() or 8))
    if args.library_gemms:
        _gemm.ENABLED = False
        if not args.dry:
            sys.path.insert(0, str(REPO / 'scripts'))
            import library_gemm_tuning as tuning        # A/B tooling outside the package
            tuning.use_tuned_gemms()
    if args.bf16:
        _gemm.PRODUCTS = 1
    micro = cfg['micro']
    # (PTMI_BENCH_UNITS: a smaller BLSTM for functional pre-flights - not the benchmark's model; the JSON line says so)
    units_override = int(os.environ['PTMI_BENCH_UNITS']) if os.environ.get('PTMI_BENCH_UNITS') else None
    model_kw = dict(units=units_override) if units_override else {}
    model = PermutationInvariantTrainingModel(**model_kw) if cfg['model'] == 'pit' else DeepClusteringModel(**model_kw)
    trainer = pt.Trainer(model, f'/tmp/ptmi_bench_{rank}', pt.optimizer.Adam(gradient_clipping=1.),
                         loss_weights=LOSS_WEIGHTS if cfg['model'] == 'pit' else None,
                         virtual_minibatch_size=world * micro, deferred_checks=not args.sync_checks,
                         overlap_allreduce=not args.no_overlap_allreduce)
    trainer.to(device)
    trainer._flat = trainer.optimizer.use_flat_grads()
    if world > 1:
        trainer._broadcast_parameters()
    hooks = trainer.enable_bucketed_allreduce()
    buckets = trainer._buckets
    model.train()
    if not args.dry:
        trainer.op_context.defer_wgrad = not args.no_overlap      # what Trainer.train() sets
        if trainer.op_context.defer_wgrad:
            _lstm.warm_side_stream(device)

    n = cfg['fs'] * SECONDS
    K = cfg['K']
    def frames_of(samples):
        return (samples + 2 * (SIZE - SHIFT) - SIZE + SHIFT - 1) // SHIFT + 1      # fading='full', pad

    lengths = None
    if args.ragged:
        import random
        rnd = random.Random(1234)                         # the same lengths on every rank: equal work per rank (weak scaling)
        lengths = sorted((rnd.randint(3 * cfg['fs'], 6 * cfg['fs']) for _ in range(cfg['batch'])), reverse=True)
        if args.row_slots:
            assert cfg['model'] == 'pit' and cfg['batch'] <= 64, 'row slots: PIT model, <= 64 slots'
            rnd = random.Random(4321)
            lengths = sorted((rnd.randint(3 * cfg['fs'], 6 * cfg['fs']) for _ in range(2 * cfg['batch'])), reverse=True)
            model.row_slots = cfg['batch']
        n = lengths[0]
    frames_per_micro = sum(frames_of(nb) for nb in lengths) if lengths else cfg['batch'] * frames_of(n)
    from padertorch_amd import _lib
    # what step() runs on: the headline's batch, or (value_ragged) a batch of SURVEY 8d's length distribution
    variant = dict(ragged=bool(args.ragged), frames=frames_per_micro, data=None)

    if args.dry:
        data = None
        nparam = trainer._flat.flat.numel()

        def step(timed, source=None):
            # stub backward: every rank's gradient = rank + 1 everywhere, announced layer by layer (last bucket first)
            buckets = trainer._buckets
            if buckets is not None and os.environ.get('PTMI_BENCH_FAKE_TIMEOUT'):
                # (tests/test_bench_launcher.py: what Trainer._check_pending raises on every rank one step after a timed-out launch)
                raise RuntimeError('padertorch_amd: a persistent LSTM kernel on cpu timed out waiting for a step counter (simulated)')
            for m in range(micro):
                if buckets is not None:
                    buckets.active = m + 1 == micro
                trainer._flat.flat.add_(float(rank + 1))
                if buckets is not None:
                    for p in reversed(trainer._flat.params):
                        buckets.ready((p,), None)
            expect = micro * world * (world + 1) / 2.
            if world > 1:
                if buckets is not None:
                    buckets.finish()
                else:
                    dist.all_reduce(trainer._flat.flat)
            got = trainer._flat.flat
            assert float(got.min()) == float(got.max()) == expect, (float(got.min()), float(got.max()), expect)
            trainer._flat.flat.zero_()
    else:
        data = synthetic_batch(1000 + rank, len(lengths) if lengths else cfg['batch'], K, n, device, lengths)
        variant['data'] = data
        timers = []

        counted = [0, 0]              # timed steps so far / of which with kernel events

        def features(src):
            feats = pt.ops.pit_features(src['y'], src['s'], src['num_samples'])
            if cfg['model'] == 'pit':
                return feats
            X = feats['X_abs'].padded                                  # [B, T, K, F]: ideal binary masks as targets
            target = torch.nn.functional.one_hot(X.argmax(2), K).permute(0, 1, 3, 2).to(torch.float32, memory_format=torch.contiguous_format)
            # (a list of per-example views of ONE padded tensor, like the features themselves: the review pads nothing)
            from padertorch_amd.ops.sequence.pack_module import PaddedList
            return dict(Y_abs=feats['Y_abs'], target_mask=PaddedList(target, feats['num_frames'], True, feats['Y_abs'].lengths_dev),
                        num_frames=feats['num_frames'])

        def step(timed, source=None):
            # the STFT feature front-end is part of the step; inside the timed region every launch of a ptmi kernel is
            # bracketed by HIP events on the stream it runs on - in every TIMER_EVERY-th step: two event packets per launch
            # are ~5 us of queue time, and a step has ~100 launches (0.5 ms per step if every step carried them)
            _lib.KERNEL_TIMERS = timers if timed and counted[0] % TIMER_EVERY == 0 else None
            if timed:
                counted[1] += counted[0] % TIMER_EVERY == 0
                counted[0] += 1
            buckets = trainer._buckets
            for m in range(micro):
                if buckets is not None:
                    buckets.active = m + 1 == micro
                src = variant['data']
                if variant['ragged']:
                    # real data brings a new length pattern every step: the per-pattern bookkeeping (ops.lstm.pack_meta: index tables,
                    # their transfers) is rebuilt every step although this bench repeats one batch
                    _lstm._meta.cache_clear()
                    from padertorch_amd.ops.sequence import slots as _slots
                    _slots._cached_layout.cache_clear()
                if source is not None:        # waveforms start in pinned host memory
                    src = dict(y=source['y'].to(device, non_blocking=True), s=source['s'].to(device, non_blocking=True),
                               num_samples=source['num_samples'])
                feats = features(src)
                assert sum(feats['num_frames']) == variant['frames'], (sum(feats['num_frames']), variant['frames'])
                loss, _, _, _ = trainer.train_step(model, feats, device)
                trainer.backward(loss)
            trainer.optimizer_step()

    def sync():
        if world > 1:
            dist.barrier()
        if not args.dry:
            torch.cuda.synchronize()

    def timed_loop(nsteps, timed=False, source=None):
        sync()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step(timed, source)
        if not args.dry:
            trainer._check_pending(flush=True)       # the last step's staged loss / grad-norm checks (deferred_checks)
        sync()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    state = dict(hooks=hooks)

    def set_overlap(flag):
        """Bucketed all-reduce under the backward pass (True) or one all-reduce of the flat buffer in optimizer_step (False)."""
        for h in state['hooks']:
            h.remove()
        trainer.op_context.grad_ready_hook = None
        trainer.op_context.grad_use_hook = None
        trainer.overlap_allreduce = bool(flag)
        state['hooks'] = trainer.enable_bucketed_allreduce()

    def recover():
        """After a recurrence watchdog timeout.  Every rank raises it in the SAME optimizer step (the timeout count travels with
        the finiteness flag through one all-reduce, Trainer.clip_grad), behind that step's collectives: the ranks are aligned."""
        trainer._pending, trainer._stage_queue, trainer._loss_acc = [], [], None
        trainer._exchanged = None
        if trainer._buckets is not None:
            trainer._buckets.reset()
        opt_ = trainer.optimizer.optimizer
        if getattr(opt_, 'found_inf', None) is not None:
            opt_.found_inf = None
        if not args.dry:
            _lstm.sync_deferred()
            torch.cuda.synchronize()
        trainer._flat.flat.zero_()

    def timed_out(e):
        return world > 1 and 'timed out' in str(e)

    # N > 1: the collectives of the bucketed all-reduce run BESIDE the persistent recurrence kernels (which need all their
    # workgroups co-resident and carry a bounded-spin watchdog).  Both schedules are probed inside the warm-up; the timed steps
    # use the faster one, and a watchdog timeout in the overlapped schedule - during the probe or the timed steps - falls back to
    # the un-overlapped one instead of ending the run.  Every decision is taken on all-reduced values: all ranks agree.
    schedule = None
    graph_state = {}

    def graph_split_step():
        """The captured data-parallel step: graph A (forward + backward of every micro-step), the 'flat+words' exchange as two RCCL
        calls, graph B (norm + clip + Adam) - train.graphed.GraphedStep with a process group."""
        from padertorch_amd.train.graphed import GraphedStep
        set_overlap(False)
        trainer.dp_protocol = 'flat+words'
        trainer._check_pending(flush=True)
        if graph_state.get('step') is None:
            graph_state['step'] = GraphedStep(trainer, [data] * micro, prepare=features, warmup=0)
        return graph_state['step']

    def graph_loop(nsteps):
        graphed = graph_state['step']
        sync()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            graphed()
        sync()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    if world > 1:
        probe = 3 if args.dry else max(3, min(10, args.steps))
        schedule = dict(probe_steps=probe, probe_ms_per_step={}, notes=[])
        if not args.dry and not args.no_graph_split and not args.eager and not args.ragged and not args.sync_checks:
            # every rank takes the same branch: the capture either works everywhere or raises everywhere (same code, same shapes); a
            # rank-local failure is agreed on through an all-reduce before anybody enters the timed collectives
            ok = 1.
            try:
                for _ in range(2):
                    step(False)
                trainer._check_pending(flush=True)
                graph_split_step()
                for _ in range(2):
                    graph_state['step']()
            except Exception as e:      # noqa
                ok = 0.
                schedule['notes'].append(f'graph_split: capture failed on rank {rank}: {type(e).__name__}: {e}')
            flag = torch.tensor([ok], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            schedule['probe_ms_per_step']['graph_split'] = None
            if float(flag.item()) == 1.:
                try:
                    schedule['probe_ms_per_step']['graph_split'] = graph_loop(probe) / probe * 1e3
                except RuntimeError as e:
                    # (a recurrence watchdog timeout is raised by EVERY rank in the same step - the count travels in the summed words -,
                    #  behind that step's collectives: the ranks are aligned and go on to the eager schedules together)
                    if not timed_out(e):
                        raise
                    recover()
                    schedule['notes'].append('graph_split: recurrence watchdog timeout during the probe')
            if schedule['probe_ms_per_step']['graph_split'] is None:
                graph_state['step'] = None
            trainer.dp_protocol = None
        for flag in ([False] if args.no_overlap_allreduce else [True, False]):
            name = 'overlap' if flag else 'no_overlap'
            set_overlap(flag)
            try:
                for _ in range(2):
                    step(False)
                schedule['probe_ms_per_step'][name] = timed_loop(probe) / probe * 1e3
            except RuntimeError as e:
                if not timed_out(e):
                    raise
                recover()
                schedule['probe_ms_per_step'][name] = None
                schedule['notes'].append(f'{name}: recurrence watchdog timeout during the probe')
        ok = {k: v for k, v in schedule['probe_ms_per_step'].items() if v is not None}
        if not ok:
            raise RuntimeError(f'bench.py: every all-reduce schedule timed out: {schedule}')
        schedule['used'] = 'overlap' if (args.dry and 'overlap' in ok) else min(ok, key=ok.get)      # (dry: stub timings mean nothing)
        set_overlap(schedule['used'] == 'overlap')
        if schedule['used'] == 'graph_split':
            graph_split_step()

    use_graph = bool(world == 1 and not args.dry and not args.eager and not args.ragged and not args.sync_checks)
    use_graph_split = bool(schedule is not None and schedule.get('used') == 'graph_split')      # (as probed; split_state['on']: as timed)
    split_state = dict(on=use_graph_split)

    def measure():
        if split_state['on']:
            graphed = graph_state['step']
            for _ in range(args.warmup):
                graphed()
            graphed.times = []                 # GPU time of graph A / the exchange / graph B per step, HIP events on the step's stream
            elapsed_graph = graph_loop(args.steps)
            graphed.times, graph_state['times'] = None, graphed.times
            # per-kernel HIP events: an eager pass of the same step (same protocol) behind the timed region, as for N = 1
            ev_steps = max(TIMER_EVERY, min(args.steps, 10 * TIMER_EVERY))
            for _ in range(2):
                step(False)
            graph_state['eager_deferred_ms'] = timed_loop(ev_steps, timed=True) / ev_steps * 1e3
            return elapsed_graph
        for _ in range(args.warmup):
            step(False)
        if not use_graph:
            return timed_loop(args.steps, timed=True)
        # N = 1: the optimizer step as ONE captured hipGraph, checks at the end of the same step (reference semantics)
        from padertorch_amd.train.graphed import GraphedStep
        trainer._check_pending(flush=True)
        if args.dp_graph and args.capture_rccl:
            trainer.graph_exchange = 'captured'          # (the layer buckets stay: their all-reduces become nodes of the graph)
        elif args.dp_graph:
            trainer.dp_protocol = 'flat+words'
            for h in state['hooks']:
                h.remove()
            state['hooks'], trainer._buckets = [], None
            trainer.op_context.grad_ready_hook = trainer.op_context.grad_use_hook = None
        # (--warmup 0 / 1: the capture still needs every lazily made table, stream and kernel attribute to exist: its own untimed eager steps)
        graphed = graph_state['step'] = GraphedStep(trainer, [data] * micro, prepare=features, warmup=max(0, 2 - args.warmup))
        if args.dp_graph and not args.capture_rccl:
            assert graphed.split
            graphed.times = []
        for _ in range(3):
            graphed()
        if args.dp_graph and not args.capture_rccl:
            graphed.times = []
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graphed()
        sync()
        elapsed_graph = time.perf_counter() - t0
        if args.dp_graph:
            graphed.times, graph_state['times'] = None, graphed.times
        # per-kernel HIP events cannot be recorded inside a replay: the same launches are bracketed in an eager pass of the same step
        # behind the timed region (rocprofv3's summary of this command sees the replays' kernels themselves: profiles/)
        if args.no_kernel_events:
            graph_state['eager_deferred_ms'] = None
            return elapsed_graph
        ev_steps = max(TIMER_EVERY, min(args.steps, 10 * TIMER_EVERY))
        for _ in range(2):
            step(False)
        graph_state['eager_deferred_ms'] = timed_loop(ev_steps, timed=True) / ev_steps * 1e3
        return elapsed_graph

    try:
        elapsed = measure()
    except RuntimeError as e:
        if not (timed_out(e) and (trainer._buckets is not None or split_state['on'])):
            raise
        recover()
        schedule['notes'].append(f"{schedule['used']}: recurrence watchdog timeout during the timed steps; rerun un-overlapped (eager)")
        schedule['used'] = 'no_overlap'
        split_state['on'] = False
        trainer.dp_protocol = None
        graph_state['step'] = None
        set_overlap(False)
        if not args.dry:
            timers.clear()
            counted[0] = counted[1] = 0
        elapsed = measure()
    hooks = state['hooks']
    buckets = trainer._buckets
    _lib.KERNEL_TIMERS = None
    frames_per_step = frames_per_micro * micro

    extras = {}
    t_trace = time.perf_counter()

    def trace(what):          # PTMI_BENCH_TRACE=1: where the wall-clock of a run goes, on stderr
        if os.environ.get('PTMI_BENCH_TRACE'):
            print(f'[bench {time.perf_counter() - t_trace:8.2f} s] {what}', file=sys.stderr, flush=True)
    trace('timed steps done')
    if not args.dry and not args.no_extras:
        nx = max(5, min(args.steps, 50))
        if use_graph:
            graphed = graph_state['step']
            extras['ms_per_step_eager_deferred'] = graph_state['eager_deferred_ms']
            trace('captured-step variants done')
        if not args.sync_checks:
            # the reference's semantics: loss and gradient norm cross to the host in the step they belong to
            opt = trainer.optimizer.optimizer
            opt.found_inf, trainer._bad = None, None
            trainer.deferred_checks = False
            for _ in range(3):
                step(False)
            extras['ms_per_step_sync_checks'] = timed_loop(nx) / nx * 1e3
            # ... and with the checks at the END of the same step (Trainer(deferred_checks='step'): device-gated update, one sync)
            trainer.deferred_checks = 'step'
            for _ in range(3):
                step(False)
            extras['ms_per_step_end_of_step_checks'] = timed_loop(nx) / nx * 1e3
            trainer.deferred_checks = True
        trace('sync-checks variant done')
        host = dict(y=data['y'].cpu().pin_memory(), s=data['s'].cpu().pin_memory(), num_samples=data['num_samples'])
        for _ in range(3):
            step(False, host)
        extras['ms_per_step_h2d'] = timed_loop(nx, source=host) / nx * 1e3
        extras['h2d_bytes_per_step'] = int((host['y'].numel() + host['s'].numel()) * 4 * micro)
        if use_graph:
            # the captured step fed from pinned host memory: the waveforms cross PCIe into the graph's static inputs at the head of the step
            graphed = graph_state['step']
            hosts = [host] * micro
            for _ in range(3):
                graphed(hosts)
            sync()
            t0 = time.perf_counter()
            for _ in range(nx):
                graphed(hosts)
            sync()
            extras['ms_per_step_graph_h2d'] = (time.perf_counter() - t0) / nx * 1e3
            graphed([data] * micro)          # (back on the resident batch)
            sync()
        # the same with the NEXT batch's transfer issued under the current step (padertorch_amd.data.DevicePrefetcher: what a
        # data pipeline in front of Trainer.train does)
        if micro == 1:
            from padertorch_amd.data import DevicePrefetcher

            def prefetched_loop(n):
                sync()
                t0 = time.perf_counter()
                for src in DevicePrefetcher((host for _ in range(n)), device):
                    feats = features(src)
                    loss, _, _, _ = trainer.train_step(model, feats, device)
                    trainer.backward(loss)
                    trainer.optimizer_step()
                trainer._check_pending(flush=True)
                sync()
                return time.perf_counter() - t0
            prefetched_loop(3)
            extras['ms_per_step_h2d_prefetched'] = prefetched_loop(nx) / nx * 1e3
            if use_graph:
                graphed = graph_state['step']

                def prefetched_graph_loop(n):
                    sync()
                    t0 = time.perf_counter()
                    # (the batch crosses PCIe one step ahead on the copy stream; its 12 MB reach the graph's static inputs device to
                    #  device BEHIND the running step's replay and in front of that step's synchronisation: `then_load`)
                    batches = iter(DevicePrefetcher((host for _ in range(n + 1)), device))
                    graphed.load([next(batches)])
                    for nxt in batches:
                        graphed(None, then_load=[nxt])
                    sync()
                    return time.perf_counter() - t0
                prefetched_graph_loop(3)
                extras['ms_per_step_graph_h2d_prefetched'] = prefetched_graph_loop(nx) / nx * 1e3
                graphed([data] * micro)
                sync()
        trace('host-to-device variants done')
        if not args.ragged and micro == 1:
            # SURVEY 8d's training distribution: lengths ~ U[3 s, 6 s] (the same draw as --ragged), zero-padded waveforms, the
            # per-pattern bookkeeping rebuilt every step as with real data
            import random
            rnd = random.Random(1234)
            rl = sorted((rnd.randint(3 * cfg['fs'], 6 * cfg['fs']) for _ in range(cfg['batch'])), reverse=True)
            variant.update(ragged=True, frames=sum(frames_of(nb) for nb in rl),
                           data=synthetic_batch(1000 + rank, cfg['batch'], K, rl[0], device, rl))
            for _ in range(3):
                step(False)
            extras['ms_per_step_ragged'] = timed_loop(nx) / nx * 1e3
            extras['value_ragged'] = variant['frames'] * world / (extras['ms_per_step_ragged'] * 1e-3)
            extras['ragged_frames_per_step'] = variant['frames'] * world
            trace('ragged variant done')
            # the same distribution on ROW SLOTS (model.row_slots; ops.sequence.SlotLayout): a recurrence costs its number of time steps,
            # not its rows, so a ragged batch only pays off with >= 2 sequences end to end per row slot - 2 x batch examples in `batch`
            # slots (occupancy 0.97 instead of 0.72); the optimizer step then covers twice the examples
            if cfg['model'] == 'pit' and cfg['batch'] <= 64:
                rnd = random.Random(4321)
                rl2 = sorted((rnd.randint(3 * cfg['fs'], 6 * cfg['fs']) for _ in range(2 * cfg['batch'])), reverse=True)
                model.row_slots = cfg['batch']
                variant.update(ragged=True, frames=sum(frames_of(nb) for nb in rl2),
                               data=synthetic_batch(2000 + rank, 2 * cfg['batch'], K, rl2[0], device, rl2))
                try:
                    for _ in range(3):
                        step(False)
                    extras['ms_per_step_ragged_row_slots'] = timed_loop(nx) / nx * 1e3
                    extras['value_ragged_row_slots'] = variant['frames'] * world / (extras['ms_per_step_ragged_row_slots'] * 1e-3)
                    extras['ragged_row_slots'] = dict(examples_per_step=2 * cfg['batch'] * world, row_slots=cfg['batch'],
                                                      frames_per_step=variant['frames'] * world)
                finally:
                    model.row_slots = None
                trace('row-slot variant done')
            variant.update(ragged=False, frames=frames_per_micro, data=data)
            if use_graph and cfg['model'] == 'pit' and cfg['batch'] <= 64:
                # the same two ragged workloads as ONE captured step each: the length pattern is device data
                # (ops.sequence.StaticSlots: a row-slot grid of fixed capacity, index tables / row masks / frame counts as tensors), so a
                # graph serves every batch that fits.  Every step brings a NEW pattern (8 draws in turn): its tables are made on the host
                # (numpy) and copied in while the GPU replays the step before.
                from padertorch_amd.ops.sequence import SlotLayout, StaticSlots
                from padertorch_amd.train.graphed import GraphedStep

                def features_slots(src):
                    feats = pt.ops.pit_features(src['y'], src['s'], src['num_samples'], num_frames_dev=src['slots'].frames)
                    return dict(feats, slots=src['slots'])

                def ragged_graph(examples, slots, seed, nbuckets=1):
                    import random
                    rnd = random.Random(seed)
                    n_max = 6 * cfg['fs']
                    T_max = frames_of(n_max)
                    pats = []
                    for p in range(8):
                        rl = sorted((rnd.randint(3 * cfg['fs'], n_max) for _ in range(examples)), reverse=True)
                        b = synthetic_batch(seed + p, examples, K, n_max, device, rl)
                        fr = [frames_of(v) for v in rl]
                        pats.append(dict(y=b['y'], s=b['s'], num_samples=torch.tensor(rl, dtype=torch.int32, device=device), frames=fr,
                                         need=SlotLayout(fr, slots).T))
                    # grid capacities (buckets): a batch takes the smallest grid it fits - one captured graph per bucket (what
                    # data.StaticSlotBatcher(steps=[...]) does in a pipeline); one example per slot: the padded length
                    needs = sorted(p['need'] for p in pats)
                    if examples == slots:
                        caps = [T_max]
                    else:
                        caps = sorted({(needs[(len(needs) * (k + 1)) // nbuckets - 1] + 7) // 8 * 8 for k in range(nbuckets)})
                    rings = {c: [StaticSlots(examples, slots, c, T_max, device) for _ in range(2)] for c in caps}
                    turn = {c: 0 for c in caps}

                    def batch(i):
                        p = pats[i % len(pats)]
                        c = next(c for c in caps if p['need'] <= c)
                        turn[c] ^= 1
                        return c, dict(y=p['y'], s=p['s'], num_samples=p['num_samples'], slots=rings[c][turn[c]].set(p['frames']))
                    trainer._check_pending(flush=True)
                    graphs = {}
                    for i in range(len(pats)):          # one capture per bucket, on the first batch that takes it
                        c, ex = batch(i)
                        if c not in graphs:
                            graphs[c] = GraphedStep(trainer, [ex], prepare=features_slots, warmup=2, clone_inputs=True)
                    state = {}

                    def prepare(i):
                        state['next'] = batch(i)

                    def loop(n):
                        prepare(0)
                        for i in range(n):
                            c, ex = state['next']
                            # the NEXT batch's tables are made on the host (numpy) and copied to its layout while this step replays
                            graphs[c]([ex], then_load=lambda i=i: prepare(i + 1))
                    loop(4)
                    sync()
                    t0 = time.perf_counter()
                    loop(nx)
                    sync()
                    ms = (time.perf_counter() - t0) / nx * 1e3
                    frames = sum(sum(pats[i % len(pats)]['frames']) for i in range(nx)) / nx
                    steps_run = sum(next(c for c in caps if pats[i % len(pats)]['need'] <= c) for i in range(nx)) / nx
                    assert all(g.captures == 1 for g in graphs.values())
                    del graphs
                    return dict(ms_per_step=ms, frames_per_step=frames * world, value=frames * world / (ms * 1e-3), examples_per_step=examples * world,
                                row_slots=slots, grid_steps=caps, mean_grid_steps=steps_run, occupancy=frames / (steps_run * slots),
                                patterns=len(pats), graphs=len(caps),
                                step_driver='one hipGraph per grid capacity for every length pattern (ops.sequence.StaticSlots), a new pattern every step')
                extras['ragged_graph'] = ragged_graph(cfg['batch'], cfg['batch'], 5000)
                extras['ms_per_step_ragged_graph'] = extras['ragged_graph']['ms_per_step']
                extras['value_ragged_graph'] = extras['ragged_graph']['value']
                trace('ragged captured step done')
                extras['ragged_row_slots_graph'] = ragged_graph(2 * cfg['batch'], cfg['batch'], 6000, nbuckets=2)
                extras['ms_per_step_ragged_row_slots_graph'] = extras['ragged_row_slots_graph']['ms_per_step']
                extras['value_ragged_row_slots_graph'] = extras['ragged_row_slots_graph']['value']
                trace('ragged row-slot captured step done')
    rccl = None
    if world > 1:
        flat = trainer._flat.flat
        reps = 2 if args.dry else 10
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        sync()
        ar_ms = (time.perf_counter() - t0) / reps * 1e3
        # every layer bucket on its own (what the overlapped schedule issues under the backward pass, last bucket first)
        bucket_ms = []
        for start, end, _ in (buckets.buckets if buckets is not None else [[0, flat.numel(), 0]]):
            seg = flat[start:end]
            sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                dist.all_reduce(seg, op=dist.ReduceOp.SUM)
            sync()
            bucket_ms.append((time.perf_counter() - t0) / reps * 1e3)
        flat.zero_()
        nbytes = flat.numel() * 4
        # the same step WITHOUT the exchange (every rank on its own gradients; the last thing the run does to the replicas): what the
        # measured step is an efficiency of, and what the two schedules predict from it
        local_ms = None
        if not args.dry:
            set_overlap(False)
            # (a measurement bracket of this script, not a switch of the Trainer: on THIS object the data-parallel branch of
            #  optimizer_step / clip_grad answers "no process group" - nothing on the product path reads a bench flag)
            trainer._dp_active = lambda: False
            try:
                for _ in range(2):
                    step(False)
                nl = max(3, min(args.steps, 20))
                local_ms = timed_loop(nl) / nl * 1e3
            finally:
                del trainer._dp_active
        # which GPU every rank is bound to (one process per GPU: the ranks must sit on DISTINCT devices - two ranks on one GPU would
        # halve the recurrences' CUs under each other and deadlock RCCL's intra-node transport)
        if args.dry:
            me = dict(rank=rank, local_rank=local_rank, device='cpu', uuid=f'cpu-{rank}', hip_visible_devices=os.environ.get('HIP_VISIBLE_DEVICES'))
        else:
            prop = torch.cuda.get_device_properties(device)
            me = dict(rank=rank, local_rank=local_rank, device=str(device), name=prop.name, uuid=str(getattr(prop, 'uuid', '')) or f'{device}',
                      pci_bus_id=getattr(prop, 'pci_bus_id', None), hip_visible_devices=os.environ.get('HIP_VISIBLE_DEVICES'),
                      rocr_visible_devices=os.environ.get('ROCR_VISIBLE_DEVICES'))
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        ids = [(r.get('uuid'), r.get('pci_bus_id')) for r in ranks]
        assert len(set(ids)) == world or os.environ.get('PTMI_BENCH_SHARE_GPU'), f'ranks share a GPU: {ranks}'
        try:
            rccl_version = '.'.join(str(v) for v in torch.cuda.nccl.version()) if not args.dry else None
        except Exception:       # (a build without the binding)
            rccl_version = None
        rccl = dict(world_size=dist.get_world_size(), backend=backend, overlap_allreduce=buckets is not None, schedule=schedule,
                    rccl_version=rccl_version, ranks=ranks,
                    # what to read the measured figures against: a ring all-reduce over xGMI moves 2 (W - 1) / W x bytes per rank and is
                    # bound by ONE link's ~153 GB/s per direction between neighbours (MI355X_MICROARCH.md); with 7 links per GPU RCCL
                    # runs several rings in parallel, so the bus bandwidth of a 94 MB buffer at W = 8 is expected between ~150 (one
                    # ring) and ~600 GB/s, i.e. 0.3 - 1.1 ms per all-reduce, under 15 % of the B = 32 step and overlappable with the
                    # backward pass (layer buckets)
                    expected=dict(bus_bandwidth_gbs=[150., 600.], blocking_all_reduce_ms_at_w8=[0.27, 1.1],
                                  weak_scaling_efficiency_floor=0.85),
                    buckets=[b[1] - b[0] for b in buckets.buckets] if buckets is not None else [flat.numel()],
                    flat_gradient_bytes=nbytes, blocking_all_reduce_ms=ar_ms, bucket_all_reduce_ms=bucket_ms,
                    # weak scaling read against THIS run's own numbers: the step without any exchange, the step as timed, and what the two
                    # schedules predict (nothing hidden: local + blocking all-reduce; everything hidden: local).  DESIGN.md section 5
                    # predicts 0.27 - 1.1 ms for the 94 MB bucket at W = 8.
                    scaling=dict(ms_per_step_without_exchange=local_ms,
                                 measured_efficiency=None if local_ms is None else local_ms / (elapsed / args.steps * 1e3),
                                 predicted_efficiency_unoverlapped=None if local_ms is None else local_ms / (local_ms + ar_ms),
                                 predicted_efficiency_fully_overlapped=None if local_ms is None else 1.0),
                    bus_bandwidth_gbs=2. * (world - 1) / world * nbytes / (ar_ms * 1e-3) / 1e9,
                    time_per_all_reduce_host_ms=trainer.timer.get('time_per_all_reduce', 0.) / max(1, trainer._opt_step) * 1e3)

    # where a rank's step time goes: wall-clock per step beside the GPU time of its parts - a first multi-GPU record
    # then separates the cost of the wire (exchange_ms: both collectives, on the step's stream) from the host's
    def part_times():
        times = graph_state.get('times') or []
        if not times:
            return None
        n = len(times)
        return dict(graph_a_ms=sum(t[0] for t in times) / n, exchange_ms=sum(t[1] for t in times) / n,
                    graph_b_ms=sum(t[2] for t in times) / n, steps=n)
    mine = dict(rank=rank, wall_ms_per_step=elapsed / args.steps * 1e3, gpu_parts=part_times(),
                host_ms_in_all_reduce_calls_per_step=trainer.timer.get('time_per_all_reduce', 0.) / max(1, trainer._opt_step) * 1e3)
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rccl is not None:
        rccl['per_rank'] = per_rank
    elif args.dp_graph and not args.dry:
        rccl = dict(world_size=1, backend=backend, schedule=dict(used='graph_captured' if args.capture_rccl else 'graph_split'), per_rank=per_rank,
                    note='one-rank process group on one GPU: everything of the captured data-parallel step but the wire')

    if rank == 0:
        out = {
            'metric': 'training frames/sec (PIT mask-est, 2-spk 8 kHz)' if args.config == 'c2' else
                      f'training frames/sec ({cfg["model"].upper()}, {K}-spk {cfg["fs"] // 1000} kHz)',
            'value': frames_per_step * world * args.steps / elapsed,
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'fp16/bf16 operands, f32 accumulate (reduced precision)' if args.bf16 else 'f32',
            'data': 'synthetic' + (f' (PRE-FLIGHT: units = {units_override}, not the benchmark model)' if units_override else ''),
            'config': {
                'workload': f'{cfg["label"]}, {(2 if args.row_slots else 1) * cfg["batch"]} x {("3-6 s (ragged, U[3 s, 6 s]" + (", end to end in " + str(cfg["batch"]) + " row slots)" if args.row_slots else ")")) if args.ragged else str(SECONDS) + " s"} {K}-spk {cfg["fs"]} Hz mixtures per GPU and '
                            f'micro-step ({frames_per_micro} frames), {micro} micro-step(s) per optimizer step, STFT '
                            f'{SIZE}/{SHIFT} on device, full optimizer step (Adam, clip 1)',
                'global_batch': cfg['batch'] * world * micro,
                'frames_per_step': frames_per_step * world,
                'parallelism': f'dp{world}',
                'blstm': 'HIP recurrence (csrc/lstm_split.hip)',
                'gemms': ('hipBLASLt/rocBLAS fp32, TunableOp selections (scripts/tuned)' if args.library_gemms else
                          'csrc/gemm_planes.hip, the hi planes only: plain 16-bit operands (reduced precision)' if args.bf16 else
                          'csrc/gemm_planes.hip: fp32 in / out, 3 16-bit MFMA products per product (fp32-equivalent accuracy); projections, '
                          'linears and weight gradients on operands pre-split into fp16 planes, LSTM input gradients on the bf16 planes the '
                          'backward recurrence hands on'),
                'host_checks': ('same step (2 syncs)' if args.sync_checks else
                                'end of the SAME optimizer step (one host synchronisation per step behind the captured step; errors raise in the '
                                'iteration they belong to, optimizer update gated on the device): train.graphed.GraphedStep'
                                if (use_graph or split_state['on']) else
                                'loss / grad-norm finiteness inspected one step late, optimizer update gated on the device (Trainer deferred_checks)'),
                'step_driver': ("two hipGraphs per optimizer step with the data-parallel exchange between them (train.graphed.GraphedStep, "
                                "split_for_allreduce: forward + backward | all_reduce(flat bucket), all_reduce(2 words) | norm + clip + Adam)"
                                if (split_state['on'] or (args.dp_graph and not args.capture_rccl)) else
                                'one hipGraph per optimizer step (train.graphed.GraphedStep), replayed' if use_graph else 'eager launches (python)'),
                'optimizer': 'csrc/optim.hip: reproducible 2-norm + fused clip / Adam / zero_grad over the flat bucket',
                'lstm_weight_gradients': 'autograd, main stream' if args.no_overlap else 'in place, side stream next to the next recurrence',
            },
        }
        if args.dry:
            out['dry'] = True
            out['roofline'] = None
        else:
            # (ragged batches: the per-kernel figures assume frames_per_step / batch time steps per launch - a reported mode without them)
            trace('extras done')
            overhead = event_bracket_overhead_ms(device)
            kernels, family = ([], None) if args.ragged else kernel_report(
                timers, max(1, counted[1]), cfg, frames_per_micro, model.blstm.hidden_size, micro, 1 if args.bf16 else 3, overhead,
                config=args.config if (use_graph and not args.bf16) else None)
            out['kernel_event_steps'] = counted[1]
            out['kernel_event_source'] = ('HIP events around the same launches in an eager pass of the same step behind the timed region (a graph '
                                          'replay takes no event records between its nodes); rocprofv3 --kernel-trace of this command times the '
                                          "replays' kernels themselves (profiles/r6_kernel_trace_bench.txt)") if (use_graph or split_state['on']) else 'HIP events inside the timed steps'
            out['event_bracket_overhead_us'] = overhead * 1e3
            # `roofline`: the ONE kernel with the most GPU time per step - what leads rocprofv3's summary of this command
            # (profiles/r4_kernel_trace_bench.txt); `roofline_family`: all planes GEMM launches together (round 3's headline entry)
            out['roofline'] = kernels[0] if kernels else None
            out['roofline_family'] = family
            out['other_kernels'] = kernels[1:]
            trace('kernel report done')
            if not args.no_extras and world == 1:
                # north_star's ">= 40 % of HBM peak on the STFT kernel": the HBM figure is the 1536-row one (1.98 GB of samples + spectra per
                # launch); the 192-row launch of rounds 2-4 fits the Infinity Cache and is labelled so
                out['other_kernels'] += standalone_front_end(device, overhead, rows=1536)
                out['other_kernels'] += standalone_front_end(device, overhead)
            trace('stand-alone front-end done')
        out.update(extras)
        # `value` is taken with the waveform batch resident in HBM (the bench contract).  SURVEY 8(d) counts example_to_device
        # inside the step: that figure, with the next batch's transfer issued one step ahead (data.DevicePrefetcher), is
        # value_to_device_inclusive; ms_per_step_h2d is the same without the prefetch (blocking copies at the head of the step)
        out['ms_per_step_resident'] = out['ms_per_step']
        if 'ms_per_step_graph_h2d_prefetched' in extras:
            out['value_to_device_inclusive'] = frames_per_step * world / (extras['ms_per_step_graph_h2d_prefetched'] * 1e-3)
        elif 'ms_per_step_h2d_prefetched' in extras:
            out['value_to_device_inclusive'] = frames_per_step * world / (extras['ms_per_step_h2d_prefetched'] * 1e-3)
        if rccl is not None:
            out['rccl'] = rccl
        if world == 1 and not args.no_cpu_baseline and not args.dry:
            out['cpu_baseline'] = cpu_baseline()
    for h in hooks:
        h.remove()
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line is the last thing on stdout (RCCL's version banner - the image exports NCCL_DEBUG=VERSION - used to follow it)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
