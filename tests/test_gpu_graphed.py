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


def _manual(path, units=48, lw=None, **kw):
    import padertorch_amd as pt
    from padertorch_amd.ops import lstm as _lstm
    m = _pit(units=units)
    t = pt.Trainer(m, path, pt.optimizer.Adam(gradient_clipping=1.), loss_weights=dict(lw or LW), deferred_checks=True, **kw)
    t.to(torch.device(DEV))
    t._flat = t.optimizer.use_flat_grads()
    t.op_context.defer_wgrad = True
    _lstm.warm_side_stream(torch.device(DEV))
    m.train()
    return m, t


def _eager_step(m, t, ex):
    loss, _, _, _ = t.train_step(m, ex, DEV)
    loss.backward()
    t.optimizer_step()


def test_replays_read_the_live_hyper_parameters_and_loss_weights(tmp_path):
    """What the reference's hooks change BETWEEN iterations reaches a replayed step: ``param_group['lr'] *= 0.5``
    (BackOffValidationHook / LRAnnealingHook, ``hooks.py:736,1029``), the clip value, betas / eps / weight decay (device words of the
    optimizer kernel), ``trainer.loss_weights`` (LossWeightAnnealingHook, ``hooks.py:957-966``: device words of the weighted sum; a
    weight crossing 0 or 1 changes the launches -> the step is captured again).  Parameters equal those of the eager loop under the
    same schedule; the summary reports the live learning rate and weights."""
    from padertorch_amd.train.graphed import GraphedStep
    exs = _examples(9, B=6)
    lw0 = dict(pit_ips_loss=1., pit_mse_loss=0.)
    (ma, ta), (mb, tb) = _manual(tmp_path / 'a', lw=lw0), _manual(tmp_path / 'b', lw=lw0)

    def schedule(t, i):
        g = t.optimizer.optimizer.param_groups[0]
        if i == 2:
            g['lr'] *= 0.5
        if i == 3:
            t.loss_weights['pit_ips_loss'] = 0.5                 # 1 -> other: new launches (a multiply and its backward)
            t.loss_weights['pit_mse_loss'] = 0.25                # 0 -> other: a new term
        if i == 4:
            t.loss_weights['pit_mse_loss'] = 0.125               # other -> other: a device word only
            t.optimizer.gradient_clipping = 0.05
        if i == 5:
            g['betas'] = (0.8, 0.99)
            g['eps'] = 1e-6
            g['weight_decay'] = 0.01
        if i == 6:
            t.loss_weights['pit_mse_loss'] = 0.                  # back to one term
            t.loss_weights['pit_ips_loss'] = 1.
    for i, ex in enumerate(exs):
        schedule(ta, i)
        _eager_step(ma, ta, ex)
    ta._check_pending(flush=True)
    _eager_step(mb, tb, exs[0])
    tb._check_pending(flush=True)
    step = GraphedStep(tb, [exs[1]], warmup=0)
    tb.train_summary.reset()
    captures = []
    for i, ex in enumerate(exs[1:], 1):
        schedule(tb, i)
        step([ex])
        captures.append(step.captures)
    assert captures == [1, 1, 2, 2, 2, 3, 3, 3], captures
    for (k, v), (_, w) in zip(ma.state_dict().items(), mb.state_dict().items()):
        np.testing.assert_allclose(w.cpu().numpy(), v.cpu().numpy(), rtol=0, atol=1e-6, err_msg=k)
    sc = tb.train_summary.data['scalars']
    assert sc['lr/param_group_0'] == [1e-3] + [5e-4] * 7, sc['lr/param_group_0']
    assert sc['pit_mse_loss_loss_weight'] == [0., 0., 0.25, 0.125, 0.125, 0., 0., 0.], sc['pit_mse_loss_loss_weight']


def test_eager_forwards_between_replays_see_the_replayed_parameters(tmp_path):
    """a replay rewrites the parameters through the graph's kernel nodes; the operand forms ``ops.gemm`` / ``ops.lstm``
    cache per parameter version for eager forwards (a validation run between two iterations, ``trainer.py:467-510``) must not stay those
    of the first such forward."""
    from padertorch_amd.train.graphed import GraphedStep
    exs = _examples(6, B=6)
    (ma, ta), (mb, tb) = _manual(tmp_path / 'a'), _manual(tmp_path / 'b')

    def validate(m):
        m.eval()
        try:
            with torch.no_grad():
                return [x.detach().clone() for x in m(exs[5])]
        finally:
            m.train()
    for ex in exs[:5]:
        _eager_step(ma, ta, ex)
    ta._check_pending(flush=True)
    want = validate(ma)
    _eager_step(mb, tb, exs[0])
    tb._check_pending(flush=True)
    step = GraphedStep(tb, [exs[1]], warmup=0)
    first = validate(mb)                          # fills the caches at the parameters' current versions
    for ex in exs[1:5]:
        step([ex])
    got = validate(mb)
    assert max(float((a - b).abs().max()) for a, b in zip(first, got)) > 1e-4         # (four steps did change the masks)
    for a, b in zip(want, got):
        torch.testing.assert_close(b, a, atol=1e-6, rtol=0)
    # ... and through the Trainer: validation at every second iteration of a graph_steps run equals the eager run's
    import padertorch_amd as pt
    records = []
    for graph in (False, True):
        m = _pit(units=48)
        t = pt.Trainer(m, tmp_path / f'v{int(graph)}', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW,
                       summary_trigger=(1, 'iteration'), checkpoint_trigger=(2, 'iteration'), stop_trigger=(6, 'iteration'),
                       graph_steps=graph)
        t.register_validation_hook(exs[4:6])
        t.train(exs[:4] * 2, device=DEV)
        records.append([s[2]['loss'] for s in t.summaries if s[1] == 'validation'])
    assert len(records[0]) == len(records[1]) >= 3
    np.testing.assert_allclose(records[1], records[0], rtol=1e-5)
    assert len(set(records[1])) == len(records[1]), records[1]


def test_signature_ignores_strings_and_trainer_reuses_the_graph(tmp_path):
    """the reference's batches carry a unique ``example_id`` per example (``pit/data.py:65``); a string is nothing a kernel
    reads, so it is no part of the signature - ``graph_steps=True`` captures once and replays."""
    import padertorch_amd as pt
    from padertorch_amd.train import graphed as G
    from padertorch_amd.train.graphed import signature
    exs = [dict(e, example_id=[f'utt{i}_{b}' for b in range(4)], dataset=f'train{i}') for i, e in enumerate(_examples(6))]
    assert signature([exs[0]]) == signature([exs[1]]) is not None
    made = []
    plain = G.GraphedStep.__init__

    def counting(self, *a, **kw):
        made.append(1)
        return plain(self, *a, **kw)
    G.GraphedStep.__init__ = counting
    try:
        t = _train(_pit(), exs, tmp_path, 6, vmb=1, graph_steps=True)
    finally:
        G.GraphedStep.__init__ = plain
    assert t.iteration == 6 and len(made) == 1, made


def test_trainer_runs_eagerly_when_a_step_cannot_be_captured(tmp_path):
    """a step whose capture raises (here: SGD, whose gradient-norm check reads the norm on the host - a synchronisation
    inside the capture) falls back to the eager step with a warning instead of ending ``train()``; same parameters as without
    ``graph_steps``, and the shape is not tried again."""
    import padertorch_amd as pt
    from padertorch_amd.train import graphed as G
    exs = _examples(5)
    out = []
    made = []
    plain = G.GraphedStep.__init__

    def counting(self, *a, **kw):
        made.append(1)
        return plain(self, *a, **kw)
    G.GraphedStep.__init__ = counting
    try:
        for graph in (False, True):
            m = _pit()
            t = pt.Trainer(m, tmp_path / str(graph), pt.optimizer.SGD(gradient_clipping=1., lr=0.05), loss_weights=LW,
                           summary_trigger=(1, 'iteration'), checkpoint_trigger=(1000, 'iteration'), stop_trigger=(5, 'iteration'),
                           graph_steps=graph)
            if graph:
                with pytest.warns(UserWarning, match='cannot be captured'):
                    t.train(exs, device=DEV)
            else:
                t.train(exs, device=DEV)
            assert t.iteration == 5
            out.append({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    finally:
        G.GraphedStep.__init__ = plain
    assert len(made) == 1, made
    for k in out[0]:
        np.testing.assert_allclose(out[1][k].numpy(), out[0][k].numpy(), rtol=0, atol=1e-6, err_msg=k)


def test_three_recurring_shapes_do_not_capture_every_step(tmp_path):
    """with more recurring shapes than graphs kept (two: a graph owns a whole step's memory) an eviction must not
    be followed by a capture at the evicted shape's next sighting - capturing costs several eager steps."""
    from padertorch_amd.train import graphed as G
    shapes = [_examples(8, N=n, seed=n) for n in (6000, 6400, 6800)]
    exs = [shapes[i % 3][i // 3] for i in range(24)]
    made = []
    plain = G.GraphedStep.__init__

    def counting(self, *a, **kw):
        made.append(1)
        return plain(self, *a, **kw)
    G.GraphedStep.__init__ = counting
    try:
        a, b = _pit(), _pit()
        ta = _train(a, exs, tmp_path / 'a', 24, vmb=1)
        tb = _train(b, exs, tmp_path / 'b', 24, vmb=1, graph_steps=True)
    finally:
        G.GraphedStep.__init__ = plain
    assert ta.iteration == tb.iteration == 24
    assert 2 <= len(made) <= 3, made
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        np.testing.assert_allclose(w.cpu().numpy(), v.cpu().numpy(), rtol=0, atol=1e-6, err_msg=k)
