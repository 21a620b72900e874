"""1500 replays of ONE captured step over ever-new length patterns (ops.sequence.StaticSlots; 64 examples of 3-6 s in 32 slots, the full
PIT model): every loss finite, no recurrence watchdog time-out, one capture."""
import os, random, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import padertorch_amd as pt
from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
from padertorch_amd.ops import lstm as L
from padertorch_amd.ops.sequence import SlotLayout, StaticSlots
from padertorch_amd.train.graphed import GraphedStep
dev = torch.device('cuda:0')
torch.manual_seed(0)
fs, B, S = 8000, int(os.environ.get('SOAK_B', 64)), int(os.environ.get('SOAK_S', 32))     # SOAK_B=32 SOAK_CAP=376 SOAK_N=12: one example per slot, for a rocprofv3 timeline
model = PermutationInvariantTrainingModel()
tr = pt.Trainer(model, '/tmp/soak_ragged', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.), deferred_checks=True)
tr.to(dev); tr._flat = tr.optimizer.use_flat_grads(); tr.op_context.defer_wgrad = True; L.warm_side_stream(dev); model.train()
stft = pt.ops.STFT(512, 128)
n_max = 6 * fs
T_max = int(stft.samples_to_frames(n_max))
g = torch.Generator().manual_seed(1)
wave = 0.1 * torch.randn(B, 2, n_max, generator=g)
rnd = random.Random(7)
def pattern():
    fixed = float(os.environ.get('SOAK_FIXED', 0))          # SOAK_FIXED=4: every example 4 s (the masked kernels on a uniform batch)
    lens = [int(fixed * fs)] * B if fixed else sorted((rnd.randint(3 * fs, n_max) for _ in range(B)), reverse=True)
    return lens, [int(stft.samples_to_frames(v)) for v in lens]
cap = int(os.environ.get('SOAK_CAP', 640))
ring = [StaticSlots(B, S, cap, T_max, dev) for _ in range(2)]
s_dev = wave.to(dev)
def batch(i):
    while True:
        lens, frames = pattern()
        if SlotLayout(frames, S).T <= cap:
            break
    s = s_dev.clone()
    ns = torch.tensor(lens, dtype=torch.int32, device=dev)
    mask = torch.arange(n_max, device=dev)[None, :] < ns[:, None]
    s = s * mask[:, None, :]
    return dict(y=s.sum(1), s=s, num_samples=ns, slots=ring[i % 2].set(frames)), sum(frames)
def features(src):
    return dict(pt.ops.pit_features(src['y'], src['s'], src['num_samples'], num_frames_dev=src['slots'].frames), slots=src['slots'])
b0, _ = batch(0)
step = GraphedStep(tr, [b0], prepare=features, warmup=2, clone_inputs=True)
N = int(os.environ.get('SOAK_N', 1500))
t0 = time.perf_counter(); frames = 0; worst = 0.
nxt = batch(1)
step.load([nxt[0]])
for i in range(N):
    cur = nxt
    holder = {}
    def make(i=i):
        holder['n'] = batch(i + 2)
        return [holder['n'][0]]
    step(None, then_load=make)
    nxt = holder['n']
    frames += cur[1]
    loss = step.scalars()['loss']
    assert loss == loss and abs(loss) < 1e6, (i, loss)
    worst = max(worst, loss)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
L.check_errors()
assert step.captures == 1
print(f'ok  ragged captured step, {N} replays over {N} length patterns (64 examples in 32 slots of {cap} steps): {dt / N * 1e3:.3f} ms/step, '
      f'{frames / dt:.0f} frames/s, last loss {loss:.4f}, largest loss {worst:.4f}, captures {step.captures}')
