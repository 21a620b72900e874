"""ONE captured optimizer step for ragged batches: the length pattern is device data (``ops.sequence.StaticSlots``),
so the graph of a ``(examples, slots, steps, padded samples)`` shape serves every batch that fits - the reference trains on
variable-length utterances (``padertorch/contrib/examples/source_separation/pit/data.py:20-33,49-77``)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
LW = dict(pit_ips_loss=1., pit_mse_loss=0.)


def _batch(seed, B, n_max, slots=None, layout=None):
    """Waveforms of lengths ~ U[n_max / 2, n_max] (sorted, zero padded to n_max) with their pattern as device data."""
    import padertorch_amd as pt
    from padertorch_amd.ops.sequence import StaticSlots
    rng = np.random.RandomState(seed)
    lens = sorted((int(v) for v in rng.randint(n_max // 2, n_max + 1, B)), reverse=True)
    g = torch.Generator().manual_seed(seed)
    s = 0.1 * torch.randn(B, 2, n_max, generator=g)
    for b, l in enumerate(lens):
        s[b, :, l:] = 0.
    stft = pt.ops.STFT(512, 128)
    frames = [int(stft.samples_to_frames(l)) for l in lens]
    T_max = int(stft.samples_to_frames(n_max))
    steps = layout if layout is not None else T_max
    st = StaticSlots(B, slots or B, steps, T_max, DEV).set(frames)
    return dict(y=s.sum(1).to(DEV), s=s.to(DEV), num_samples=torch.tensor(lens, dtype=torch.int32, device=DEV), slots=st), lens, frames


def _pit(units=48, layers=2):
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    torch.manual_seed(0)
    return PermutationInvariantTrainingModel(F=257, recurrent_layers=layers, units=units, K=2)


def _trainer(model, path):
    import padertorch_amd as pt
    from padertorch_amd.ops import lstm as _lstm
    t = pt.Trainer(model, path, pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW, deferred_checks=True)
    t.to(torch.device(DEV))
    t._flat = t.optimizer.use_flat_grads()
    t.op_context.defer_wgrad = True
    _lstm.warm_side_stream(torch.device(DEV))
    model.train()
    return t


def _features(src):
    import padertorch_amd as pt
    feats = pt.ops.pit_features(src['y'], src['s'], src['num_samples'], num_frames_dev=src['slots'].frames)
    return dict(feats, slots=src['slots'])


@pytest.mark.parametrize('B,slots,steps,units', [(8, 8, None, 48), (24, 16, 96, 100), (64, 32, 120, 600)])
def test_one_graph_serves_changing_length_patterns(tmp_path, B, slots, steps, units):
    """Six batches with six different length patterns through ONE graph (``captures == 1``): per-step losses, gradient norms and the
    parameters after six Adam steps equal the eager loop's over the same batches - on the same layout (1e-6) and, for the losses, on
    the PackedSequence path the reference's batches take (one sequence per row, 1e-5)."""
    from padertorch_amd.train.graphed import GraphedStep, signature
    import padertorch_amd as pt
    n_max = 6400
    batches = [_batch(100 + i, B, n_max, slots, steps) for i in range(7)]
    assert len({tuple(l) for _, l, _ in batches}) == 7
    assert len({signature([b]) for b, _, _ in batches}) == 1            # one signature: shapes only
    (ma, mb, mc) = (_pit(units), _pit(units), _pit(units))
    ta, tb, tc = _trainer(ma, tmp_path / 'a'), _trainer(mb, tmp_path / 'b'), _trainer(mc, tmp_path / 'c')
    losses = {0: [], 1: [], 2: []}
    for src, lens, _ in batches:                                          # eager, the same layout
        loss, _, _, _ = ta.train_step(ma, _features(src), DEV)
        loss.backward()
        losses[0].append(float(loss.detach()))
        ta.optimizer_step()
    ta._check_pending(flush=True)
    for src, lens, _ in batches:                                          # eager, PackedSequence (host-side lengths)
        loss, _, _, _ = tc.train_step(mc, pt.ops.pit_features(src['y'], src['s'], lens), DEV)
        loss.backward()
        losses[2].append(float(loss.detach()))
        tc.optimizer_step()
    tc._check_pending(flush=True)
    # the graph: batch 0 eagerly (warm-up, a real step), then replays on batches 1 .. 6
    loss, _, _, _ = tb.train_step(mb, _features(batches[0][0]), DEV)
    loss.backward()
    losses[1].append(float(loss.detach()))
    tb.optimizer_step()
    tb._check_pending(flush=True)
    step = GraphedStep(tb, [batches[1][0]], prepare=_features, warmup=0, clone_inputs=True)
    for src, _, _ in batches[1:]:
        step([src])
        losses[1].append(step.scalars()['loss'])
    assert step.captures == 1
    np.testing.assert_allclose(losses[1], losses[0], rtol=1e-6)
    np.testing.assert_allclose(losses[1][:2], losses[2][:2], rtol=1e-5)
    np.testing.assert_allclose(losses[1], losses[2], rtol=2e-3)          # (two fp32 trajectories under Adam)
    for (k, v), (_, w) in zip(ma.state_dict().items(), mb.state_dict().items()):
        np.testing.assert_allclose(w.cpu().numpy(), v.cpu().numpy(), rtol=0, atol=1e-6, err_msg=k)


def test_a_batch_that_does_not_fit_is_refused():
    from padertorch_amd.ops.sequence import StaticSlots
    st = StaticSlots(4, 2, 30, 20, DEV)
    st.set([20, 15, 10, 5])
    with pytest.raises(ValueError):
        st.set([20, 20, 20, 20])                 # 40 steps per slot
    with pytest.raises(ValueError):
        st.set([21, 5, 5, 5])                    # longer than the padded time
    with pytest.raises(ValueError):
        st.set([5, 5, 5])
    assert st.fits([20, 10, 10, 9]) and not st.fits([20, 20, 20, 20])


def test_dc_model_on_static_slots_equals_row_slots_and_shares_one_graph(tmp_path):
    """The deep-clustering model (``contrib/tcl/dc.py``; BASELINE configs[4]'s model) on the device-data layout: embeddings, loss and every
    gradient equal those of the host-side row-slot layout (``model.row_slots``: itself held against the oracle in
    ``tests/test_gpu_fullsize.py``), and five length patterns replay through one captured step with the eager loop's parameters."""
    import padertorch_amd as pt
    from padertorch_amd.contrib.tcl.dc import DeepClusteringModel
    from padertorch_amd.ops.sequence.pack_module import PaddedList
    from padertorch_amd.train.graphed import GraphedStep
    B, S, n_max, K = 24, 16, 6400, 3

    def dc_batch(seed):
        src, lens, frames = _batch(seed, B, n_max, S, 96)
        g = torch.Generator().manual_seed(seed + 1)
        s3 = 0.1 * torch.randn(B, K, n_max, generator=g)
        for b, l in enumerate(lens):
            s3[b, :, l:] = 0.
        return dict(y=s3.sum(1).to(DEV), s=s3.to(DEV), num_samples=src['num_samples'], slots=src['slots']), lens, frames

    def features(src, host_lens=None):
        if host_lens is None:
            feats = pt.ops.pit_features(src['y'], src['s'], src['num_samples'], num_frames_dev=src['slots'].frames)
        else:
            feats = pt.ops.pit_features(src['y'], src['s'], host_lens)
        X = feats['X_abs'].padded
        target = torch.nn.functional.one_hot(X.argmax(2), K).permute(0, 1, 3, 2).to(torch.float32, memory_format=torch.contiguous_format)
        out = dict(Y_abs=feats['Y_abs'], target_mask=PaddedList(target, feats['Y_abs'].lengths, True, feats['Y_abs'].lengths_dev))
        if host_lens is None:
            out['slots'] = src['slots']
        return out

    def model():
        torch.manual_seed(2)
        return DeepClusteringModel(units=64, recurrent_layers=2, E=8, input_feature_transform='log1p')
    src, lens, frames = dc_batch(300)
    ma, mb = model().to(DEV).train(), model().to(DEV).train()
    mb.row_slots = S
    fa, fb = features(src), features(src, lens)
    ea, eb = ma(fa), mb(fb)
    for b, t in enumerate(frames):
        torch.testing.assert_close(ea[b][:t], eb[b], atol=1e-6, rtol=0)
        assert float(ea[b][t:].abs().sum()) == 0.
    la, lb = ma.review(fa, ea)['losses']['dc_loss'], mb.review(fb, eb)['losses']['dc_loss']
    assert abs(float(la) - float(lb)) < 1e-6 * max(1., abs(float(lb)))
    la.backward()
    lb.backward()
    for (k, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
        scale = float(q.grad.abs().max())
        assert float((p.grad - q.grad).abs().max()) <= 1e-5 * scale + 1e-12, k
    # one graph, five patterns
    batches = [dc_batch(310 + i)[0] for i in range(6)]
    m1, m2 = model(), model()

    def trainer(m, path):
        t = pt.Trainer(m, path, pt.optimizer.Adam(gradient_clipping=1.), deferred_checks=True)
        t.to(torch.device(DEV))
        t._flat = t.optimizer.use_flat_grads()
        t.op_context.defer_wgrad = True
        m.train()
        return t
    t1, t2 = trainer(m1, tmp_path / '1'), trainer(m2, tmp_path / '2')
    for srcb in batches:
        loss, _, _, _ = t1.train_step(m1, features(srcb), DEV)
        loss.backward()
        t1.optimizer_step()
    t1._check_pending(flush=True)
    loss, _, _, _ = t2.train_step(m2, features(batches[0]), DEV)
    loss.backward()
    t2.optimizer_step()
    t2._check_pending(flush=True)
    step = GraphedStep(t2, [batches[1]], prepare=features, warmup=0, clone_inputs=True)
    for srcb in batches[1:]:
        step([srcb])
    assert step.captures == 1
    for (k, v), (_, w) in zip(m1.state_dict().items(), m2.state_dict().items()):
        np.testing.assert_allclose(w.cpu().numpy(), v.cpu().numpy(), rtol=0, atol=1e-6, err_msg=k)


def test_trainer_trains_a_ragged_stream_on_one_graph(tmp_path):
    """The user's path end to end: a stream of variable-length utterances (``pit/data.py:20-33``) -> ``data.row_slot_batches`` ->
    ``data.StaticSlotBatcher`` -> ``Trainer.train(graph_steps=True)``.  ``example_to_device`` makes the features from the device-side
    lengths, the first batch runs eagerly, the second captures, all later ones replay through that ONE graph although every batch has
    its own length pattern; a batch that does not fit the grid runs eagerly in between.  Parameters equal those of the eager loop."""
    import padertorch_amd as pt
    from padertorch_amd.data import StaticSlotBatcher, row_slot_batches
    from padertorch_amd.train import graphed as G
    rng = np.random.RandomState(4)
    lens = [int(v) for v in rng.randint(3200, 6401, 8 * 7)]
    lens[8 * 3:8 * 4] = [6400] * 8                                  # the fourth batch is too long for the grid: handed back, eager
    stream = [dict(y=(0.1 * rng.randn(n)).astype(np.float32), s=(0.1 * rng.randn(2, n)).astype(np.float32), num_samples=n, example_id=f'u{i}')
              for i, n in enumerate(lens)]

    def run(graph, path):
        batcher = StaticSlotBatcher(examples=8, slots=4, max_samples=6400, device=DEV, steps=96)
        data = [batcher(b) for b in row_slot_batches(stream, row_slots=4, fill=2.0)]
        assert batcher.refused == 1 and 'slots' not in data[3]
        model = _pit(48)
        model.row_slots = 4                                          # (the route of a batch that was handed back)
        made = []
        plain = G.GraphedStep.__init__
        G.GraphedStep.__init__ = lambda self, *a, **k: (made.append(1), plain(self, *a, **k))[1]
        try:
            t = pt.Trainer(model, path, pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW, summary_trigger=(1, 'iteration'),
                           checkpoint_trigger=(1000, 'iteration'), stop_trigger=(7, 'iteration'), graph_steps=graph)
            t.train(data, device=DEV)
        finally:
            G.GraphedStep.__init__ = plain
        return model, t, len(made)
    ma, ta, _ = run(False, tmp_path / 'a')
    mb, tb, captures = run(True, tmp_path / 'b')
    assert ta.iteration == tb.iteration == 7 and captures == 1, captures
    for (k, v), (_, w) in zip(ma.state_dict().items(), mb.state_dict().items()):
        np.testing.assert_allclose(w.cpu().numpy(), v.cpu().numpy(), rtol=0, atol=1e-6, err_msg=k)
    la = [s[2]['loss'] for s in ta.summaries if s[1] == 'training']
    lb = [s[2]['loss'] for s in tb.summaries if s[1] == 'training']
    np.testing.assert_allclose(lb, la, rtol=1e-5)
