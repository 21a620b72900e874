"""SURVEY.md section 8b, "Threading" (reference ``padertorch/train/trainer.py:412-420``: one model replica per device, each
driven from its own host thread): the ops of this package are re-entrant per model and per stream.  Round 4 retired the
process-global switches the Trainer used to set around ``train()`` (``ops.lstm.DEFER_WGRAD`` / ``GRAD_*_HOOK`` -> a per-model
``ops.context.OpContext``; ``ops.lstm.LAST_HANDOFF`` -> a record that travels with the tensor).  Two different models trained
alternately in one process, and two models trained concurrently from two host threads on two streams, end bit-identical to the
same models trained alone."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _examples(seed, lens, K):
    from oracle import features_np
    rng = np.random.RandomState(seed)
    return [features_np.synthetic_mixture(rng, n, K=K) for n in lens]          # [(s [K, n], y [n])]


def _make(kind, tmp, tag):
    """(trainer, batch): two DIFFERENT models - sizes, K, loss weights, the in-place weight-gradient switch, clipping."""
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    if kind == 'a':
        torch.manual_seed(1)
        model = PermutationInvariantTrainingModel(F=257, recurrent_layers=2, units=24, K=2)
        exs = _examples(5, [4000, 4000, 4000, 4000], 2)
        t = pt.Trainer(model, tmp / f'{tag}_a', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.))
        defer = True
    else:
        torch.manual_seed(2)
        model = PermutationInvariantTrainingModel(F=257, recurrent_layers=1, units=16, K=3)
        exs = _examples(6, [3600, 3000, 2200], 3)
        t = pt.Trainer(model, tmp / f'{tag}_b', pt.optimizer.Adam(gradient_clipping=5.), loss_weights=dict(pit_ips_loss=.5, pit_mse_loss=.5))
        defer = False
    t.to(DEV)
    t.optimizer.use_flat_grads()
    t.op_context.defer_wgrad = defer                 # what Trainer.train() sets for this model only
    batch = dict(y=[torch.from_numpy(y).to(DEV) for _, y in exs], s=[torch.from_numpy(s).to(DEV) for s, _ in exs])
    return t, batch


def _step(t, batch):
    loss, _, _, _ = t.train_step(t.model, batch, DEV)
    loss.backward()
    t.optimizer_step()
    return loss.detach()


def _alone(kind, tmp, steps=3):
    t, batch = _make(kind, tmp, 'alone')
    losses = [_step(t, batch) for _ in range(steps)]
    t._check_pending(flush=True)
    torch.cuda.synchronize()
    return [float(v) for v in losses], {k: v.clone() for k, v in t.model.state_dict().items()}


def test_two_models_trained_alternately_equal_the_single_model_runs(tmp_path):
    from padertorch_amd.ops import lstm as L
    ref = {k: _alone(k, tmp_path) for k in 'ab'}
    ta, ba = _make('a', tmp_path, 'alt')
    tb, bb = _make('b', tmp_path, 'alt')
    assert ta.op_context is not tb.op_context and ta.op_context.defer_wgrad and not tb.op_context.defer_wgrad
    la, lb = [], []
    for _ in range(3):
        la.append(_step(ta, ba))
        lb.append(_step(tb, bb))
    for t in (ta, tb):
        t._check_pending(flush=True)
    torch.cuda.synchronize()
    assert (L.DEFER_WGRAD, L.GRAD_READY_HOOK, L.GRAD_USE_HOOK) == (False, None, None)          # no process-global was touched
    assert not hasattr(L, 'LAST_HANDOFF')
    for t, losses, key in ((ta, la, 'a'), (tb, lb, 'b')):
        assert [float(v) for v in losses] == ref[key][0], key
        for k, v in t.model.state_dict().items():
            assert torch.equal(v, ref[key][1][k]), (key, k)


def test_two_host_threads_on_two_streams_equal_the_single_model_runs(tmp_path):
    from padertorch_amd.ops import lstm as L
    ref = {k: _alone(k, tmp_path) for k in 'ab'}
    out, errors = {}, []
    barrier = threading.Barrier(2)
    build = threading.Lock()             # torch's global RNG seeds the models: one construction at a time

    def work(kind):
        try:
            torch.cuda.set_device(DEV)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                with build:
                    t, batch = _make(kind, tmp_path, 'thr')
                    L.warm_side_stream(DEV)          # this stream's own weight-gradient side stream
                barrier.wait()
                losses = []
                for _ in range(3):
                    losses.append(_step(t, batch))
                t._check_pending(flush=True)
                stream.synchronize()
                out[kind] = ([float(v) for v in losses], {k: v.clone() for k, v in t.model.state_dict().items()})
        except BaseException as e:       # noqa: BLE001  (reported by the main thread)
            errors.append((kind, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=(k,)) for k in 'ab']
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    torch.cuda.synchronize()
    assert not errors, errors
    L.check_errors()
    for key in 'ab':
        assert out[key][0] == ref[key][0], key
        for k, v in out[key][1].items():
            assert torch.equal(v, ref[key][1][k]), (key, k)
