"""The bench step of BASELINE configs[1] (or --config c3 / c5) eager against captured (train.graphed.GraphedStep):
same losses from the same state, ms per step in the four modes (eager deferred / eager 'step' / graph 'step' / graph deferred).

    python scripts/exp_graph_step.py [--config c2] [--steps 60]
"""
import argparse
import json
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

import torch  # noqa: E402

import bench  # noqa: E402
import padertorch_amd as pt  # noqa: E402
from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel  # noqa: E402
from padertorch_amd.contrib.tcl.dc import DeepClusteringModel  # noqa: E402
from padertorch_amd.ops import gemm as _gemm, lstm as _lstm  # noqa: E402
from padertorch_amd.train.graphed import GraphedStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2')
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--parity-steps', type=int, default=4)
    ap.add_argument('--replay-only', type=int, default=0, help='capture, then only replay this many steps (for rocprofv3 --kernel-trace)')
    ap.add_argument('--eager-only', type=int, default=0, help='only this many eager steps, deferred checks (for rocprofv3 --kernel-trace)')
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    device = torch.device('cuda', 0)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = PermutationInvariantTrainingModel() if cfg['model'] == 'pit' else DeepClusteringModel()
    micro = cfg['micro']
    trainer = pt.Trainer(model, '/tmp/ptmi_exp_graph', pt.optimizer.Adam(gradient_clipping=1.),
                         loss_weights=bench.LOSS_WEIGHTS if cfg['model'] == 'pit' else None, virtual_minibatch_size=micro,
                         deferred_checks=True)
    trainer.to(device)
    trainer._flat = trainer.optimizer.use_flat_grads()
    model.train()
    trainer.op_context.defer_wgrad = True
    _lstm.warm_side_stream(device)
    n = cfg['fs'] * bench.SECONDS
    K = cfg['K']
    datas = [bench.synthetic_batch(1000 + m, cfg['batch'], K, n, device) for m in range(micro)]

    def features(src):
        feats = pt.ops.pit_features(src['y'], src['s'], src['num_samples'])
        if cfg['model'] == 'pit':
            return feats
        X = feats['X_abs'].padded
        target = torch.nn.functional.one_hot(X.argmax(2), K).permute(0, 1, 3, 2).to(torch.float32, memory_format=torch.contiguous_format)
        from padertorch_amd.ops.sequence.pack_module import PaddedList
        return dict(Y_abs=feats['Y_abs'], target_mask=PaddedList(target, feats['num_frames'], True, feats['Y_abs'].lengths_dev),
                    num_frames=feats['num_frames'])

    losses = []

    def eager_step():
        for m in range(micro):
            loss, _, _, _ = trainer.train_step(model, features(datas[m]), device)
            loss.backward()
            losses.append(loss.detach())
        trainer.optimizer_step()

    def timed(fn, nsteps, finish=None):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            fn()
        if finish is not None:
            finish()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / nsteps * 1e3

    out = {'config': args.config}
    if args.replay_only:
        step = GraphedStep(trainer, datas, prepare=features)
        ms = timed(step, args.replay_only)
        print(json.dumps({'config': args.config, 'graph_step_checks_ms': ms}), flush=True)
        return
    if args.eager_only:
        for _ in range(3):
            eager_step()
        ms = timed(eager_step, args.eager_only, lambda: trainer._check_pending(flush=True))
        print(json.dumps({'config': args.config, 'eager_deferred_ms': ms}), flush=True)
        return
    for _ in range(5):
        eager_step()
    trainer._check_pending(flush=True)
    trainer.deferred_checks = True
    out['eager_deferred_ms'] = timed(eager_step, args.steps, lambda: trainer._check_pending(flush=True))
    trainer.deferred_checks = 'step'
    for _ in range(3):
        eager_step()
    out['eager_step_checks_ms'] = timed(eager_step, args.steps)
    trainer.deferred_checks = True
    print(json.dumps(out), flush=True)

    # the state both paths start the parity run from
    torch.cuda.synchronize()
    snap_p = [p.detach().clone() for p in trainer._flat.params]

    def restore():
        torch.cuda.synchronize()
        with torch.no_grad():
            for p, q in zip(trainer._flat.params, snap_p):
                p.copy_(q)
            opt = trainer.optimizer
            if opt._bound is not None:
                opt._bound[0].zero_()
                opt._bound[1].zero_()
                opt._bound[2].zero_()
            trainer._flat.flat.zero_()
        _gemm.invalidate()
        torch.cuda.synchronize()

    restore()
    del losses[:]
    for _ in range(args.parity_steps):
        eager_step()
    trainer._check_pending(flush=True)
    torch.cuda.synchronize()
    want = [float(v) for v in losses]
    want_p = [p.detach().clone() for p in trainer._flat.params]

    t0 = time.perf_counter()
    step = GraphedStep(trainer, datas, prepare=features)
    out['capture_s'] = time.perf_counter() - t0
    restore()
    got = []
    for _ in range(args.parity_steps):
        step()
        got.append(step.scalars()['loss'])
    torch.cuda.synchronize()
    # (one loss per optimizer step is staged last: compare with the last micro-step's)
    want_last = want[micro - 1::micro]
    out['loss_eager'] = want_last
    out['loss_graph'] = got
    out['loss_max_rel_diff'] = max(abs(a - b) / max(abs(a), 1e-12) for a, b in zip(want_last, got))
    out['param_max_abs_diff'] = max(float((p.detach() - q).abs().max()) for p, q in zip(trainer._flat.params, want_p))
    print(json.dumps(out), flush=True)

    for _ in range(5):
        step()
    out['graph_step_checks_ms'] = timed(step, args.steps)
    # with fresh data copied into the static inputs every step (what a training loop does)
    fresh = [dict(y=d['y'].clone(), s=d['s'].clone(), num_samples=d['num_samples']) for d in datas]
    out['graph_step_checks_with_input_copy_ms'] = timed(lambda: step(fresh), args.steps)
    print(json.dumps(out), flush=True)
    # a non-finite input: raises in the same call, parameters untouched
    before = [p.detach().clone() for p in trainer._flat.params]
    bad = [dict(y=d['y'].clone(), s=d['s'].clone(), num_samples=d['num_samples']) for d in datas]
    bad[0]['y'][0, 100] = float('nan')
    try:
        step(bad)
        out['nan_raises'] = False
    except RuntimeError as e:
        out['nan_raises'] = 'not finite' in str(e)
    out['nan_scalars'] = {k: repr(v) for k, v in step.scalars().items()}
    out['nan_static_input_has_nan'] = bool(torch.isnan(step._inputs[0][0]).any())
    torch.cuda.synchronize()
    out['nan_leaves_parameters'] = all(torch.equal(p.detach(), q) for p, q in zip(trainer._flat.params, before))
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
