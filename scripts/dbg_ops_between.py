"""Which torch op launches every small kernel of the c2 step (eager, deferred checks): kernel name <- innermost CPU op, in launch order.

    python scripts/dbg_ops_between.py [--from pit_pairwise --to lstm_bwd]
"""
import argparse
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

import torch  # noqa: E402

import bench  # noqa: E402
import padertorch_amd as pt  # noqa: E402
from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel  # noqa: E402
from padertorch_amd.ops import lstm as _lstm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--all', action='store_true')
    args = ap.parse_args()
    cfg = bench.CONFIGS['c2']
    device = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = PermutationInvariantTrainingModel()
    trainer = pt.Trainer(model, '/tmp/ptmi_dbg_ops', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=bench.LOSS_WEIGHTS,
                         deferred_checks=True)
    trainer.to(device)
    trainer._flat = trainer.optimizer.use_flat_grads()
    model.train()
    trainer.op_context.defer_wgrad = True
    _lstm.warm_side_stream(device)
    data = bench.synthetic_batch(1000, cfg['batch'], cfg['K'], cfg['fs'] * bench.SECONDS, device)

    def step():
        feats = pt.ops.pit_features(data['y'], data['s'], data['num_samples'])
        loss, _, _, _ = trainer.train_step(model, feats, device)
        loss.backward()
        trainer.optimizer_step()

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    evs = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU), key=lambda e: (e.time_range.start, -e.time_range.end))
    t0 = evs[0].time_range.start
    lo = next(e.time_range.start for e in evs if 'pit_loss_forward' in e.name) - 50
    hi = [e.time_range.end for e in evs if 'relu_backward' in e.name][0] + 20
    stack = []
    for e in evs:
        if not args.all and not (lo <= e.time_range.start <= hi):
            continue
        while stack and stack[-1] < e.time_range.end and stack[-1] <= e.time_range.start:
            stack.pop()
        ks = ', '.join(f'{k.name[:40]} {k.duration:.1f}us' for k in e.kernels)
        print(f'{e.time_range.start - t0:9.1f} {"  " * len(stack)}{e.name[:90]}   {"[" + ks + "]" if ks else ""}')
        stack.append(e.time_range.end)


if __name__ == '__main__':
    main()
