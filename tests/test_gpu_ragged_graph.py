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
