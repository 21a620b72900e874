#!/usr/bin/env python3
"""Row-slot batches: host and device time of one training step's pieces (debugging aid)."""
import random
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import padertorch_amd as pt  # noqa: E402
from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel  # noqa: E402
from padertorch_amd.ops.sequence import SlotLayout, slots as _slots  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = PermutationInvariantTrainingModel().to(dev).train()
rnd = random.Random(4321)
n_ex, S = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 32
lens = sorted((rnd.randint(24000, 48000) for _ in range(n_ex)), reverse=True)
g = torch.Generator().manual_seed(1)
s = 0.1 * torch.randn(n_ex, 2, lens[0], generator=g)
for b, l in enumerate(lens):
    s[b, :, l:] = 0
s = s.to(dev)
y = s.sum(1)
model.row_slots = S


def tick(label, t0):
    torch.cuda.synchronize()
    print(f'{label:28s} {1e3 * (time.perf_counter() - t0):9.2f} ms', flush=True)
    return time.perf_counter()


for it in range(3):
    print('--- iteration', it)
    t0 = time.perf_counter()
    _slots._cached_layout.cache_clear()
    feats = pt.ops.pit_features(y, s, lens)
    t0 = tick('features', t0)
    L = SlotLayout.cached(tuple(feats['num_frames']), S, dev)
    L.meta
    t0 = tick(f'layout T={L.T} occ={L.occupancy:.2f}', t0)
    masks = model(feats)
    t0 = tick('forward', t0)
    loss = model.review(feats, masks)['losses']['pit_ips_loss']
    t0 = tick('review', t0)
    loss.backward()
    t0 = tick('backward', t0)
    pt.ops.lstm.check_errors()

# the Trainer's path: in-place weight gradients on the side stream, flat gradient bucket, fused optimizer
print('=== Trainer path')
trainer = pt.Trainer(model, '/tmp/ptmi_exp_slots', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.),
                     deferred_checks=True)
trainer.to(dev)
trainer._flat = trainer.optimizer.use_flat_grads()
trainer.op_context.defer_wgrad = True
pt.ops.lstm.warm_side_stream(dev)
import faulthandler
faulthandler.dump_traceback_later(40, exit=True)
for it in range(6):
    t0 = time.perf_counter()
    _slots._cached_layout.cache_clear()
    pt.ops.lstm._meta.cache_clear()
    feats = pt.ops.pit_features(y, s, lens)
    loss, _, _, _ = trainer.train_step(model, feats, dev)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    trainer.optimizer_step()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f'step {it}: host forward {1e3 * (t1 - t0):.1f} backward {1e3 * (t2 - t1):.1f} optimizer {1e3 * (t3 - t2):.1f} drain {1e3 * (t4 - t3):.1f} ms', flush=True)
trainer._check_pending(flush=True)
