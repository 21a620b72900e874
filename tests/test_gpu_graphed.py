"""The captured optimizer step (``train.graphed.GraphedStep`` / ``Trainer(graph_steps=True)``: one hipGraph per step) against the eager
step it captures: same parameters, same losses, same error behaviour (reference loop: ``padertorch/train/trainer.py:357-393,512-565``)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
LW = dict(pit_ips_loss=1., pit_mse_loss=0.)


def _pit(seed=0, **kw):
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    torch.manual_seed(seed)
    return PermutationInvariantTrainingModel(**dict(dict(F=257, recurrent_layers=2, units=32, K=2), **kw))


def _examples(n, B=4, N=6000, seed=0):
    """Feature batches of ONE shape (equal lengths): what a graph can be shared by."""
    from padertorch_amd.ops import pit_features
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        s = 0.1 * torch.randn(B, 2, N, generator=g)
        f = pit_features(s.sum(1).to(DEV), s.to(DEV))
        out.append({k: (list(v) if isinstance(v, list) else v) for k, v in f.items()})      # plain lists: the reference's batch contract
    torch.cuda.synchronize()
    return out


def _train(model, exs, tmp, steps, vmb=2, **kw):
    import padertorch_amd as pt
    t = pt.Trainer(model, tmp, pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW, summary_trigger=(1, 'iteration'),
                   checkpoint_trigger=(1000, 'iteration'), stop_trigger=(steps, 'iteration'), virtual_minibatch_size=vmb, **kw)
    t.train(exs, device=DEV)
    return t


@pytest.mark.parametrize('deferred', [False, True, 'step'])
def test_trainer_graph_steps_train_like_the_eager_loop(tmp_path, deferred):
    """Six optimizer steps of two micro-steps each over examples of one shape: the first step of the shape runs eagerly, the second
    captures, the rest replay - parameters, per-iteration losses and gradient norms equal those of the eager loop."""
    exs = _examples(12)
    a, b = _pit(), _pit()
    ta = _train(a, exs, tmp_path / 'a', 6, deferred_checks=deferred)
    tb = _train(b, exs, tmp_path / 'b', 6, deferred_checks=deferred, graph_steps=True)
    assert ta.iteration == tb.iteration == 6
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        np.testing.assert_allclose(w.cpu().numpy(), v.cpu().numpy(), rtol=0, atol=1e-6, err_msg=k)
    sa = [s[2] for s in ta.summaries if s[1] == 'training']
    sb = [s[2] for s in tb.summaries if s[1] == 'training']
    assert len(sa) == len(sb) >= 5
    for x, y in zip(sa, sb):
        assert set(x) == set(y), (sorted(x), sorted(y))
        for key in x:
            np.testing.assert_allclose(y[key], x[key], rtol=1e-5, atol=1e-7, err_msg=key)


@pytest.mark.parametrize('deferred', [False, True, 'step'])
def test_trainer_graph_steps_non_finite_loss(tmp_path, deferred):
    """A NaN in the fifth optimizer step's input (a REPLAYED step): the reference's RuntimeError in the same iteration - also with
    deferred_checks=True, whose one-step delay only the eager steps have -, the parameters those after step four (the update is gated on
    the device)."""
    exs = _examples(12)
    ref = _pit()
    _train(ref, exs, tmp_path / 'a', 4)
    bad = [dict(e) for e in exs]
    bad[8] = dict(bad[8], X_abs=[x * float('nan') for x in bad[8]['X_abs']])
    model = _pit()
    import padertorch_amd as pt
    t = pt.Trainer(model, tmp_path / 'b', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW, summary_trigger=(1000, 'iteration'),
                   checkpoint_trigger=(1000, 'iteration'), stop_trigger=(6, 'iteration'), virtual_minibatch_size=2,
                   deferred_checks=deferred, graph_steps=True)
    with pytest.raises(RuntimeError, match='is not finite'):
        t.train(bad, device=DEV)
    assert t.iteration == 4, t.iteration
    for (k, v), (_, r) in zip(model.state_dict().items(), ref.state_dict().items()):
        np.testing.assert_allclose(v.cpu().numpy(), r.cpu().numpy(), rtol=0, atol=1e-6, err_msg=k)


def test_graphed_step_with_the_feature_front_end_inside(tmp_path):
    """``GraphedStep(prepare=pit_features)`` on raw waveforms (what ``bench.py`` times): new waveforms are copied into the static inputs,
    losses and parameters follow the eager step bit for bit over changing batches; ``then_load`` stages the next batch behind the replay."""
    import padertorch_amd as pt
    from padertorch_amd.train.graphed import GraphedStep
    g = torch.Generator().manual_seed(3)
    waves = []
    for _ in range(5):
        s = 0.1 * torch.randn(8, 2, 8000, generator=g)
        waves.append(dict(y=s.sum(1).to(DEV), s=s.to(DEV)))

    def features(src):
        return pt.ops.pit_features(src['y'], src['s'])

    def make(path):
        m = _pit(units=64)
        t = pt.Trainer(m, path, pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW, deferred_checks=True)
        t.to(torch.device(DEV))
        t._flat = t.optimizer.use_flat_grads()
        t.op_context.defer_wgrad = True
        m.train()
        return m, t
    ma, ta = make(tmp_path / 'a')
    mb, tb = make(tmp_path / 'b')
    losses_a = []
    for w in waves:
        loss, _, _, _ = ta.train_step(ma, features(w), DEV)
        loss.backward()
        losses_a.append(float(loss))
        ta.optimizer_step()
    ta._check_pending(flush=True)
    # the graph: warm-up step on batch 0 (a real step), then replays over batches 1 .. 4
    loss, _, _, _ = tb.train_step(mb, features(waves[0]), DEV)
    loss.backward()
    tb.optimizer_step()
    tb._check_pending(flush=True)
    static = dict(y=waves[1]['y'].clone(), s=waves[1]['s'].clone())
    step = GraphedStep(tb, [static], prepare=features, warmup=0)
    losses_b = [float(loss)]
    step()
    losses_b.append(step.scalars()['loss'])
    step([waves[2]])
    losses_b.append(step.scalars()['loss'])
    step.load([waves[3]])
    step(None, then_load=[waves[4]])
    losses_b.append(step.scalars()['loss'])
    step()
    losses_b.append(step.scalars()['loss'])
    np.testing.assert_allclose(losses_b, losses_a, rtol=1e-6)
    for (k, v), (_, w) in zip(ma.state_dict().items(), mb.state_dict().items()):
        np.testing.assert_allclose(w.cpu().numpy(), v.cpu().numpy(), rtol=0, atol=1e-6, err_msg=k)
    assert float(waves[1]['y'].abs().sum()) > 0            # (the callers' tensors are not the static inputs)


def test_signature_tells_shapes_lengths_and_unknown_leaves_apart():
    from padertorch_amd.ops.sequence.pack_module import PaddedList
    from padertorch_amd.train.graphed import signature
    a = dict(x=torch.zeros(3, 4, device=DEV), n=[4, 4, 3])
    assert signature([a]) == signature([dict(x=torch.ones(3, 4, device=DEV), n=[4, 4, 3])])
    assert signature([a]) != signature([dict(x=torch.zeros(3, 5, device=DEV), n=[4, 4, 3])])
    assert signature([a]) != signature([dict(x=torch.zeros(3, 4, device=DEV), n=[4, 4, 2])])
    pl = PaddedList(torch.zeros(2, 5, 3, device=DEV), [5, 4])
    assert signature([dict(y=pl)]) != signature([dict(y=PaddedList(torch.zeros(2, 5, 3, device=DEV), [5, 3]))])
    assert signature([dict(x=object())]) is None


def test_trainer_graph_prepare_captures_the_feature_front_end(tmp_path):
    """``Trainer.graph_prepare``: raw waveform examples go through ``Trainer.train(graph_steps=True)``, the feature front-end is part of
    the captured step; same parameters as the eager loop over features made in the data pipeline."""
    import padertorch_amd as pt
    g = torch.Generator().manual_seed(11)
    waves = []
    for _ in range(6):
        s = 0.1 * torch.randn(8, 2, 6000, generator=g)
        waves.append(dict(y=s.sum(1).to(DEV), s=s.to(DEV)))

    def features(src):
        return pt.ops.pit_features(src['y'], src['s'])
    a, b = _pit(units=48), _pit(units=48)
    ta = _train(a, (features(w) for w in waves), tmp_path / 'a', 6, vmb=1)
    tb = pt.Trainer(b, tmp_path / 'b', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW, summary_trigger=(1, 'iteration'),
                    checkpoint_trigger=(1000, 'iteration'), stop_trigger=(6, 'iteration'), graph_steps=True)
    tb.graph_prepare = features
    tb.train(waves, device=DEV)
    assert ta.iteration == tb.iteration == 6 and len(tb._graphs) == 0          # (train() lets its graphs go at the end)
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        np.testing.assert_allclose(w.cpu().numpy(), v.cpu().numpy(), rtol=0, atol=1e-6, err_msg=k)
