"""Eager step, single process: gradients with the dense layers' weight gradients deferred to the recurrence launch (ops.linear.DEFER_IN_EAGER)
against not deferred - flat gradient buffers compared after one and after two micro-steps."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import padertorch_amd as pt
import bench
from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
from padertorch_amd.ops import linear as LL, lstm as L
dev = torch.device('cuda', 0)
toy = len(sys.argv) > 1 and sys.argv[1] == 'toy'
KW = dict(F=257, recurrent_layers=2, units=32, K=2) if toy else {}
BN = (4, 6000) if toy else (32, 32000)
outs = {}
for mode in (False, True, False, True):
    LL.DEFER_IN_EAGER = mode
    torch.manual_seed(0)
    model = PermutationInvariantTrainingModel(**KW)
    tr = pt.Trainer(model, '/tmp/dbg_defer', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=bench.LOSS_WEIGHTS, deferred_checks=True)
    tr.to(dev); tr._flat = tr.optimizer.use_flat_grads(); model.train(); tr.op_context.defer_wgrad = True
    L.warm_side_stream(dev)
    res = []
    for m in range(2):
        d = bench.synthetic_batch(1000 + m, BN[0], 2, BN[1], dev)
        feats = pt.ops.pit_features(d['y'], d['s'], d['num_samples'])
        loss, _, _, _ = tr.train_step(model, feats, dev)
        tr.backward(loss)
        L.sync_deferred(dev)
        torch.cuda.synchronize()
        res.append(tr._flat.flat.clone())
    outs.setdefault(mode, []).append(res)
    tr.optimizer_step()
    del tr, model
for k in range(2):
    a, b = outs[False][0][k], outs[True][0][k]
    a2, b2 = outs[False][1][k], outs[True][1][k]
    print(f'micro-step {k}: deferred vs not: max |diff| {float((a - b).abs().max()):.3e} of {float(a.abs().max()):.3e}; run-to-run not deferred {float((a - a2).abs().max()):.3e}, deferred {float((b - b2).abs().max()):.3e}')
    # where
    names = [(n, p.numel()) for n, p in PermutationInvariantTrainingModel(**KW).named_parameters()]
    off = 0
    for n, c in names:
        d = float((a[off:off + c] - b[off:off + c]).abs().max())
        if d > 0:
            print(f'    {n}: {d:.3e}')
        off += c
