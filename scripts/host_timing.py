#!/usr/bin/env python3
"""How far ahead of the GPU is the host in the bench step?  Per step: host time spent enqueuing (step start -> the
deferred check of the previous step), time blocked in that check (0 = the host is the bottleneck), host time per phase.
    python scripts/host_timing.py [c2|c3] [ragged]   (ragged: lengths U[3 s, 6 s], the per-pattern bookkeeping rebuilt every step)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import padertorch_amd as pt
from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
from padertorch_amd.ops import lstm as _lstm
import bench

dev = torch.device('cuda', 0)
torch.manual_seed(0)
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in bench.CONFIGS else 'c2']
RAGGED = 'ragged' in sys.argv
model = PermutationInvariantTrainingModel()
trainer = pt.Trainer(model, '/tmp/ptmi_ht', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=bench.LOSS_WEIGHTS,
                     virtual_minibatch_size=1, deferred_checks=True)
trainer.to(dev)
trainer._flat = trainer.optimizer.use_flat_grads()
model.train()
_lstm.DEFER_WGRAD = True
_lstm.warm_side_stream(dev)
n = cfg['fs'] * bench.SECONDS
lengths = None
if RAGGED:
    import random
    rnd = random.Random(1234)
    lengths = sorted((rnd.randint(3 * cfg['fs'], 6 * cfg['fs']) for _ in range(cfg['batch'])), reverse=True)
    n = lengths[0]
data = bench.synthetic_batch(1000, cfg['batch'], cfg['K'], n, dev, lengths)

marks = {}
def add(k, dt):
    marks[k] = marks.get(k, 0.) + dt

orig_check = trainer._check_pending
def timed_check(flush=False):
    t = time.perf_counter(); orig_check(flush); add('blocked_in_check', time.perf_counter() - t)
trainer._check_pending = timed_check

def step():
    if RAGGED:
        _lstm._meta.cache_clear()
    t0 = time.perf_counter()
    feats = pt.ops.pit_features(data['y'], data['s'], data['num_samples'])
    t1 = time.perf_counter()
    loss, _, _, _ = trainer.train_step(model, feats, dev)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    trainer.optimizer_step()
    t4 = time.perf_counter()
    add('features', t1 - t0); add('forward+review', t2 - t1); add('backward', t3 - t2); add('optimizer_step', t4 - t3)

for _ in range(10):
    step()
torch.cuda.synchronize()
marks.clear()
N = 100
t0 = time.perf_counter()
for _ in range(N):
    step()
trainer._check_pending(flush=True)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f'ms/step {el / N * 1e3:.3f}')
for k, v in marks.items():
    print(f'host {k:18s} {v / N * 1e3:7.3f} ms/step')
