"""Does the host turn-around between two captured steps (~80 us: synchronise, inspect, hipGraphLaunch) go away when the next replay is
enqueued before the last one is checked?  One graph exec relaunched without synchronisation against TWO execs (two captures of the same
step, sharing parameters / moments / gradients) launched alternately.

    python scripts/exp_graph_pipeline.py [--config c2] [--steps 60]
"""
import argparse
import json
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / 'scripts'))

import torch  # noqa: E402

import bench  # noqa: E402
import padertorch_amd as pt  # noqa: E402
from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel  # noqa: E402
from padertorch_amd.ops import lstm as _lstm  # noqa: E402
from padertorch_amd.train.graphed import GraphedStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2')
    ap.add_argument('--steps', type=int, default=60)
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    device = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = PermutationInvariantTrainingModel()
    micro = cfg['micro']
    trainer = pt.Trainer(model, '/tmp/ptmi_exp_pipe', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=bench.LOSS_WEIGHTS,
                         virtual_minibatch_size=micro, deferred_checks=True)
    trainer.to(device)
    trainer._flat = trainer.optimizer.use_flat_grads()
    model.train()
    trainer.op_context.defer_wgrad = True
    _lstm.warm_side_stream(device)
    n = cfg['fs'] * bench.SECONDS
    datas = [bench.synthetic_batch(1000 + m, cfg['batch'], cfg['K'], n, device) for m in range(micro)]

    def features(src):
        return pt.ops.pit_features(src['y'], src['s'], src['num_samples'])

    a = GraphedStep(trainer, datas, prepare=features)
    b = GraphedStep(trainer, datas, prepare=features)
    stream = torch.cuda.current_stream(device)

    def timed(fn, nsteps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nsteps):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / nsteps * 1e3

    out = {'config': args.config}
    for rep in range(2):
        out[f'one_exec_checked_every_step_ms_{rep}'] = timed(lambda i: a(), args.steps)
        out[f'one_exec_no_sync_ms_{rep}'] = timed(lambda i: a._graph.replay(), args.steps)
        out[f'two_execs_no_sync_ms_{rep}'] = timed(lambda i: (a if i % 2 == 0 else b)._graph.replay(), args.steps)

        def alt_sync(i):
            (a if i % 2 == 0 else b)._graph.replay()
            stream.synchronize()
        out[f'two_execs_sync_ms_{rep}'] = timed(alt_sync, args.steps)

        def one_behind(i, ev=[None, None]):
            # launch step i, then wait for step i - 1 (its event), i.e. the host is one step ahead
            (a if i % 2 == 0 else b)._graph.replay()
            e = torch.cuda.Event()
            e.record(stream)
            if ev[0] is not None:
                ev[0].synchronize()
            ev[0] = e
        out[f'two_execs_checked_one_behind_ms_{rep}'] = timed(one_behind, args.steps)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
