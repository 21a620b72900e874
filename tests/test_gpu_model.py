"""PIT / DC models + Trainer on the GPU vs the reference goldens (G6) and the torch-CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _batch(g6, keys):
    Ts = [int(t) for t in g6['Ts']]
    b = {k: [g6[f'in_{k}_{i}'] for i in range(len(Ts))] for k in keys}
    b['num_frames'] = Ts
    return b


def _load(model, g6, prefix):
    model.load_state_dict({k[len(prefix):]: torch.from_numpy(v) for k, v in g6.items()
                           if isinstance(v, np.ndarray) and k.startswith(prefix)}, strict=True)
    return model


def test_pit_model_vs_reference(g6):
    """Reference state_dict loads (identical keys/shapes); masks, both losses and all gradients match;
    minibatch loss == mean of single-example losses (tests/test_models/test_bss.py:153-192)."""
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    model = _load(PermutationInvariantTrainingModel(F=9, recurrent_layers=2, units=4, K=2), g6, 'pit_sd_').to(DEV)
    batch = model.example_to_device(_batch(g6, ['Y_abs', 'X_abs', 'cos_phase_difference']), DEV)
    masks = model(batch)
    assert isinstance(masks, list) and len(masks) == 3
    for b, m in enumerate(masks):
        assert m.shape == g6[f'pit_mask_{b}'].shape
        np.testing.assert_allclose(m.detach().cpu().numpy(), g6[f'pit_mask_{b}'], atol=1e-5)
    rv = model.review(batch, masks)
    assert set(rv) == {'losses'} and set(rv['losses']) == {'pit_mse_loss', 'pit_ips_loss'}
    np.testing.assert_allclose(rv['losses']['pit_mse_loss'].item(), g6['pit_mse_loss'], atol=1e-5)
    np.testing.assert_allclose(rv['losses']['pit_ips_loss'].item(), g6['pit_ips_loss'], atol=1e-5)
    rv['losses']['pit_ips_loss'].backward()
    for n, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), g6[f'pit_grad_{n}'], atol=1e-5, err_msg=n)
    singles = []
    for b in range(3):
        ex = {k: [v[b]] for k, v in batch.items()}
        r = model.review(ex, model(ex))['losses']
        singles.append([r['pit_mse_loss'].item(), r['pit_ips_loss'].item()])
    np.testing.assert_allclose(np.mean(singles, 0), [g6['pit_mse_loss'], g6['pit_ips_loss']], atol=1e-5)
    np.testing.assert_allclose(singles, g6['pit_single_losses'], atol=1e-5)
    # snapshot images only on request
    model.create_snapshot = True
    rv = model.review(batch, model(batch))
    assert set(rv['images']) == {'observation', 'mask_0', 'mask_1', 'estimation_0', 'estimation_1'}


def test_trainer_three_steps_vs_reference(g6, tmp_path):
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    model = _load(PermutationInvariantTrainingModel(F=9, recurrent_layers=2, units=4, K=2), g6, 'pit_sd_')
    batch = _batch(g6, ['Y_abs', 'X_abs', 'cos_phase_difference'])
    exs = [{k: [v[b] for b in idx] for k, v in batch.items()} for idx in g6['train_example_indices']]
    t = pt.Trainer(model, tmp_path, pt.optimizer.Adam(gradient_clipping=1.),
                   loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.), summary_trigger=(1000, 'iteration'),
                   checkpoint_trigger=(1000, 'iteration'), stop_trigger=(3, 'iteration'), virtual_minibatch_size=2)
    t.train(exs, device=DEV)
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g6['pit_sd3_' + k], atol=2e-5, err_msg=k)


def test_trainer_deferred_checks_match_golden(g6, tmp_path):
    """deferred_checks=True (loss / grad norm inspected one step late, update gated on the device) trains
    to the same parameters as the reference's three optimizer steps."""
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    model = _load(PermutationInvariantTrainingModel(F=9, recurrent_layers=2, units=4, K=2), g6, 'pit_sd_')
    batch = _batch(g6, ['Y_abs', 'X_abs', 'cos_phase_difference'])
    exs = [{k: [v[b] for b in idx] for k, v in batch.items()} for idx in g6['train_example_indices']]
    t = pt.Trainer(model, tmp_path, pt.optimizer.Adam(gradient_clipping=1.),
                   loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.), summary_trigger=(1000, 'iteration'),
                   checkpoint_trigger=(1000, 'iteration'), stop_trigger=(3, 'iteration'), virtual_minibatch_size=2,
                   deferred_checks=True)
    t.train(exs, device=DEV)
    assert not t._pending
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g6['pit_sd3_' + k], atol=2e-5, err_msg=k)
    scalars = t.summaries[-1][2]
    assert np.isfinite(scalars['loss']) and np.isfinite(scalars['grad_norm']), scalars


def test_host_to_device_staging_wraps_without_losing_values(monkeypatch):
    """``_lib.host_to_device``: small host arrays through two pinned halves, many laps, two streams: every device tensor holds its values
    (a half is rewritten only behind the event of the copies out of it); large arrays take their own pinned buffer."""
    from padertorch_amd import _lib
    monkeypatch.setattr(_lib, '_STAGING', {0: _lib._Staging(half_bytes=1 << 12)})
    rng = np.random.RandomState(0)
    side = torch.cuda.Stream()
    keep = []
    for i in range(400):
        a = rng.randint(-1000, 1000, size=rng.randint(1, 120)).astype(np.int64)
        if i % 3 == 0:
            with torch.cuda.stream(side):
                keep.append((a, _lib.host_to_device(a, torch.int64, DEV)))
        else:
            keep.append((a, _lib.host_to_device(a.tolist(), torch.int32, torch.device(DEV))))
    big = rng.randint(0, 9, size=5000).astype(np.int32)
    keep.append((big, _lib.host_to_device(big, torch.int32, DEV)))
    torch.cuda.synchronize()
    for a, d in keep:
        assert d.is_cuda and d.cpu().numpy().tolist() == a.tolist()
    assert _lib.host_to_device([], torch.int32, DEV).numel() == 0


def test_unpack_sequence_ragged_is_one_scatter_and_matches_torch():
    """``ops.unpack_sequence`` of a ragged PackedSequence on the GPU (reference ``pack_module.py:29-30``): values, zero padding, lengths
    and the gradient equal ``torch.nn.utils.rnn.pad_packed_sequence``'s."""
    from torch.nn.utils.rnn import pack_sequence, pad_packed_sequence
    from padertorch_amd import ops
    torch.manual_seed(5)
    lens = [37, 30, 30, 12, 5, 1]
    seqs = [torch.randn(n, 2, 7, device=DEV) for n in lens]
    a = pack_sequence(seqs)
    data1 = a.data.clone().requires_grad_(True)
    data2 = a.data.clone().requires_grad_(True)
    out = ops.unpack_sequence(a._replace(data=data1))
    ref, ref_lens = pad_packed_sequence(a._replace(data=data2))
    assert out.lengths == ref_lens.tolist() == lens and not out.batch_first
    assert torch.equal(out.padded, ref)
    for b, n in enumerate(lens):
        assert torch.equal(out[b], seqs[b])
    w = torch.randn_like(ref)
    (out.padded * w).sum().backward()
    (ref * w).sum().backward()
    assert torch.equal(data1.grad, data2.grad)


def test_trainer_prefetched_weight_forms_change_nothing(g6, tmp_path, monkeypatch):
    """From the second optimizer step on, the dense layers' operand forms (maximum, planes) and the later LSTM layers' are made on the
    preparation stream behind the optimizer kernel (ops.gemm.note_update / prefetch_known, ops.lstm._stacked_weights): the cache
    entries carry that stream's event, and training ends at the very same parameters as with everything made at first use."""
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    from padertorch_amd.ops import gemm as G
    batch = _batch(g6, ['Y_abs', 'X_abs', 'cos_phase_difference'])
    exs = [{k: [v[b] for b in idx] for k, v in batch.items()} for idx in g6['train_example_indices']]

    def run(sub):
        G.invalidate()
        model = _load(PermutationInvariantTrainingModel(F=9, recurrent_layers=2, units=4, K=2), g6, 'pit_sd_')
        t = pt.Trainer(model, tmp_path / sub, pt.optimizer.Adam(gradient_clipping=1.),
                       loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.), summary_trigger=(1000, 'iteration'),
                       checkpoint_trigger=(1000, 'iteration'), stop_trigger=(3, 'iteration'), virtual_minibatch_size=2,
                       deferred_checks=True)
        t.train(exs, device=DEV)
        return model

    model = run('a')
    w1 = model.linear1.weight
    assert id(w1) in G._KNOWN and G._KNOWN[id(w1)][0]() is w1 and G._KNOWN[id(w1)][1]
    # (the last forward pass ran on forms the preparation stream had made: an event is attached)
    entries = [e for k, e in G._WEIGHT_PLANES.items() if (k == id(w1) or (isinstance(k, tuple) and id(w1) in k)) and len(e) == 5]
    assert entries and any(e[4] is not None for e in entries), [len(e) for e in entries]
    want = {k: v.clone() for k, v in model.state_dict().items()}
    monkeypatch.setattr(G, 'prefetch_known', lambda device: None)
    monkeypatch.setattr(G, 'update_event', lambda params: None)
    got = run('b').state_dict()
    for k, v in want.items():
        assert torch.equal(v, got[k]), k


@pytest.mark.parametrize('deferred', [False, True, 'step'])
def test_trainer_non_finite_loss_raises_and_keeps_parameters(g6, tmp_path, deferred):
    """A NaN in the third optimizer step's input: RuntimeError('The loss (nan) is not finite...') as in
    trainer.py:620-638, the parameters are those after step two (the deferred mode skips the update on
    the device and raises one iteration late; 'step': device-gated too, but inspected at the end of the SAME optimizer step)."""
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    batch = _batch(g6, ['Y_abs', 'X_abs', 'cos_phase_difference'])
    exs = [{k: [v[b] for b in idx] for k, v in batch.items()} for idx in g6['train_example_indices']]
    exs = (exs + exs)[:8]
    kw = dict(loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.), summary_trigger=(1000, 'iteration'),
              checkpoint_trigger=(1000, 'iteration'), virtual_minibatch_size=2)
    ref = _load(PermutationInvariantTrainingModel(F=9, recurrent_layers=2, units=4, K=2), g6, 'pit_sd_')
    pt.Trainer(ref, tmp_path / 'a', pt.optimizer.Adam(gradient_clipping=1.), stop_trigger=(2, 'iteration'),
               **kw).train(exs, device=DEV)
    bad = [dict(e) for e in exs]
    bad[4] = dict(bad[4], X_abs=[x * float('nan') for x in bad[4]['X_abs']])
    model = _load(PermutationInvariantTrainingModel(F=9, recurrent_layers=2, units=4, K=2), g6, 'pit_sd_')
    t = pt.Trainer(model, tmp_path / 'b', pt.optimizer.Adam(gradient_clipping=1.), stop_trigger=(4, 'iteration'),
                   deferred_checks=deferred, **kw)
    with pytest.raises(RuntimeError, match='is not finite'):
        t.train(bad, device=DEV)
    assert t.iteration == (3 if deferred is True else 2), t.iteration    # optimizer steps gone through (deferred: the third one skipped on the device)
    for (k, v), (_, r) in zip(model.state_dict().items(), ref.state_dict().items()):
        np.testing.assert_array_equal(v.cpu().numpy(), r.cpu().numpy(), err_msg=k)


def test_trainer_test_run_on_gpu(g6, tmp_path):
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    model = _load(PermutationInvariantTrainingModel(F=9, recurrent_layers=2, units=4, K=2), g6, 'pit_sd_')
    batch = _batch(g6, ['Y_abs', 'X_abs', 'cos_phase_difference'])
    exs = [{k: [v[b] for b in idx] for k, v in batch.items()} for idx in g6['train_example_indices']]
    t = pt.Trainer(model, tmp_path, pt.optimizer.Adam(gradient_clipping=1.),
                   loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.), virtual_minibatch_size=2)
    t.test_run(exs, exs[:2], device=DEV)


def test_dc_model_vs_reference(g6):
    from padertorch_amd.contrib.tcl.dc import DeepClusteringModel
    model = _load(DeepClusteringModel(F=9, recurrent_layers=1, units=4, E=3), g6, 'dc_sd_').to(DEV)
    batch = model.example_to_device(_batch(g6, ['Y_abs', 'target_mask']), DEV)
    emb = model(batch)
    for b, m in enumerate(emb):
        np.testing.assert_allclose(m.detach().cpu().numpy(), g6[f'dc_emb_{b}'], atol=1e-5)
    rv = model.review(batch, emb)
    np.testing.assert_allclose(rv['losses']['dc_loss'].item(), g6['dc_loss'], atol=1e-5)
    rv['losses']['dc_loss'].backward()
    for n, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), g6[f'dc_grad_{n}'], atol=1e-5, err_msg=n)


def test_dc_loss_vs_reference(g1, g5):
    from padertorch_amd.ops import deep_clustering_loss
    for toy in g1['dc_toys']:
        got = deep_clustering_loss(torch.tensor(toy['embedding'], dtype=torch.float32, device=DEV),
                                   torch.tensor(toy['target'], dtype=torch.float32, device=DEV))
        np.testing.assert_allclose(got.item(), toy['loss'], atol=1e-6)
    x = torch.from_numpy(g5['x']).to(DEV).requires_grad_(True)
    loss = deep_clustering_loss(x, torch.from_numpy(g5['t']).to(DEV))
    np.testing.assert_allclose(loss.item(), g5['loss'], atol=1e-5)
    loss.backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g5['grad'], atol=1e-6)


@pytest.mark.parametrize('N,E,K', [(5000, 40, 3), (257 * 60, 31, 2), (300, 64, 5), (17, 33, 1)])
def test_dc_loss_wider_than_one_matrix_core_tile_vs_oracle(N, E, K):
    """E + K > 32 (the reference has no such limit, ``source_separation.py:13-31``): the three products and the gradient on the planes
    GEMM - value against the fp64 oracle, gradient against torch autograd of the reference formula in fp64; and a DeepClusteringModel
    with such an embedding width reviews through it (strict: nothing leaves the hand-written path)."""
    from oracle import losses_np
    from padertorch_amd.ops import deep_clustering_loss
    rng = np.random.RandomState(N + E)
    x = rng.standard_normal((N, E)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    t = np.eye(K, dtype=np.float32)[rng.randint(0, K, N)]
    xd = torch.from_numpy(x).to(DEV).requires_grad_(True)
    loss = deep_clustering_loss(xd, torch.from_numpy(t).to(DEV))
    want = losses_np.deep_clustering_loss(x, t)
    np.testing.assert_allclose(loss.item(), want, rtol=2e-6, atol=1e-7)
    loss.backward()
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    t64 = torch.from_numpy(t).double()
    ref = (torch.sum((x64.t() @ x64) ** 2) - 2 * torch.sum((x64.t() @ t64) ** 2) + torch.sum((t64.t() @ t64) ** 2)) / N ** 2
    ref.backward()
    np.testing.assert_allclose(xd.grad.cpu().numpy(), x64.grad.numpy(), rtol=0, atol=2e-6 * float(x64.grad.abs().max()))


def test_dc_model_with_a_wide_embedding_reviews_on_the_hip_path():
    from oracle import torch_ref
    from padertorch_amd.contrib.tcl.dc import DeepClusteringModel
    torch.manual_seed(0)
    kw = dict(F=33, recurrent_layers=1, units=16, E=36)
    model = DeepClusteringModel(**kw)
    ref = torch_ref.DCModelRef(**kw)
    ref.load_state_dict(model.state_dict())
    model.to(DEV).train()
    lens = [40, 31, 12]
    Y = [torch.rand(n, 33) for n in lens]
    tm = [torch.nn.functional.one_hot(torch.randint(0, 3, (n, 33)), 3).permute(0, 2, 1).float() for n in lens]
    batch = dict(Y_abs=[y.to(DEV) for y in Y], target_mask=[m.to(DEV) for m in tm])
    loss = model.review(batch, model(batch))['losses']['dc_loss']
    rb = dict(Y_abs=Y, target_mask=tm)
    rloss = ref.review(rb, ref(rb))['losses']['dc_loss']
    assert abs(float(loss) - float(rloss)) < 1e-5, (float(loss), float(rloss))
    loss.backward()
    rloss.backward()
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.numpy(), atol=2e-4 * float(q.grad.abs().max()) + 1e-9, err_msg=n)


def test_dc_loss_batched_full_size_vs_oracle():
    """Config-5 shape (K=3, E=20, F=257): fused batched kernel on the model's (T,B,E,F) layout vs the
    fp64 oracle per example, value and gradient; ragged lengths."""
    from oracle import losses_np
    from padertorch_amd.ops.losses import dc_loss_batched
    rng = np.random.RandomState(5)
    lens, E, K, F = [37, 30, 30, 11], 20, 3, 257
    B, T = len(lens), max(lens)
    emb = rng.standard_normal((T, B, E, F)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=2, keepdims=True)
    tm = np.eye(K, dtype=np.float32)[rng.randint(0, K, (B, T, F))].transpose(0, 1, 3, 2).copy()
    ref = [losses_np.deep_clustering_loss(emb[:l, b].transpose(0, 2, 1).reshape(-1, E),
                                          tm[b, :l].transpose(0, 2, 1).reshape(-1, K)) for b, l in enumerate(lens)]
    x = torch.from_numpy(emb).to(DEV).requires_grad_(True)
    ld = torch.tensor(lens, dtype=torch.int32, device=DEV)
    loss, ex = dc_loss_batched(x, torch.from_numpy(tm).to(DEV), ld)
    np.testing.assert_allclose(ex.cpu().numpy(), ref, rtol=2e-5)
    np.testing.assert_allclose(loss.item(), np.mean(ref), rtol=2e-5)
    loss.backward()
    xt = torch.from_numpy(emb).double().requires_grad_(True)
    tot = 0
    for b, l in enumerate(lens):
        X = xt[:l, b].permute(0, 2, 1).reshape(-1, E)
        Tm = torch.from_numpy(tm[b, :l]).double().permute(0, 2, 1).reshape(-1, K)
        N = X.shape[0]
        tot = tot + (((X.t() @ X) ** 2).sum() - 2 * ((X.t() @ Tm) ** 2).sum() + ((Tm.t() @ Tm) ** 2).sum()) / N ** 2 / B
    tot.backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), xt.grad.numpy(), atol=1e-9, rtol=2e-4)


@pytest.mark.parametrize('E,K,F,lens', [
    (24, 8, 40, [9, 9, 4]),        # the limits of the branch-free buffer-load kernels (E <= 24, K <= 8)
    (25, 3, 33, [7, 5]),           # one past: generic kernels
    (1, 1, 257, [3]),
    (8, 2, 5, [20, 20, 20, 1]),    # rows of a tile spanning many time steps
    (16, 4, 64, [6] * 5),          # equal lengths: no row_frames
])
def test_dc_loss_batched_shapes_vs_oracle(E, K, F, lens):
    """dc_loss_batched on the model's (T, B, E, F) layout across both kernel families (fast: buffer loads
    with out-of-range offsets for absent rows / columns; generic), value and gradient vs the fp64 oracle."""
    from oracle import losses_np
    from padertorch_amd.ops.losses import dc_loss_batched
    rng = np.random.RandomState(E * 100 + K)
    B, T = len(lens), max(lens)
    emb = rng.standard_normal((T, B, E, F)).astype(np.float32)
    tm = np.eye(K, dtype=np.float32)[rng.randint(0, K, (B, T, F))].transpose(0, 1, 3, 2).copy()
    ref = [losses_np.deep_clustering_loss(emb[:l, b].transpose(0, 2, 1).reshape(-1, E),
                                          tm[b, :l].transpose(0, 2, 1).reshape(-1, K)) for b, l in enumerate(lens)]
    x = torch.from_numpy(emb).to(DEV).requires_grad_(True)
    ld = None if len(set(lens)) == 1 else torch.tensor(lens, dtype=torch.int32, device=DEV)
    loss, ex = dc_loss_batched(x, torch.from_numpy(tm).to(DEV), ld)
    np.testing.assert_allclose(ex.cpu().numpy(), ref, rtol=3e-5)
    loss.backward()
    xt = torch.from_numpy(emb).double().requires_grad_(True)
    tot = 0
    for b, l in enumerate(lens):
        X = xt[:l, b].permute(0, 2, 1).reshape(-1, E)
        Tm = torch.from_numpy(tm[b, :l]).double().permute(0, 2, 1).reshape(-1, K)
        N = X.shape[0]
        tot = tot + (((X.t() @ X) ** 2).sum() - 2 * ((X.t() @ Tm) ** 2).sum() + ((Tm.t() @ Tm) ** 2).sum()) / N ** 2 / B
    tot.backward()
    g = x.grad.cpu().numpy()
    np.testing.assert_allclose(g, xt.grad.numpy(), atol=1e-7 * max(1.0, float(np.abs(xt.grad.numpy()).max())), rtol=3e-4)
    for b, l in enumerate(lens):
        assert not g[l:, b].any()                      # padded frames get exactly zero


def test_dc_loss_full_size_properties():
    """C5 frame count per example (503 x 257 bins, E=20, K=3; 16 examples): the loss only sees X X^T and
    T T^T, so it is invariant under a rotation of the embedding axis (with a covariant gradient) and under
    a permutation of the target classes; the fused batch result is the mean of the per-example results."""
    from padertorch_amd.ops.losses import dc_loss_batched
    torch.manual_seed(2)
    T, B, E, K, F = 503, 16, 20, 3, 257
    x = torch.nn.functional.normalize(torch.randn(T, B, E, F, device=DEV), dim=2).requires_grad_(True)
    tm = torch.nn.functional.one_hot(torch.randint(0, K, (B, T, F), device=DEV), K).permute(0, 1, 3, 2).float().contiguous()
    loss, ex = dc_loss_batched(x, tm)
    loss.backward()
    np.testing.assert_allclose(loss.item(), ex.mean().item(), rtol=1e-6)
    q, _ = torch.linalg.qr(torch.randn(E, E, device=DEV))
    x2 = torch.einsum('tbef,eg->tbgf', x.detach(), q).contiguous().requires_grad_(True)
    loss2, ex2 = dc_loss_batched(x2, tm[:, :, [2, 0, 1]].contiguous())
    loss2.backward()
    np.testing.assert_allclose(ex2.cpu().numpy(), ex.detach().cpu().numpy(), rtol=2e-4)
    g_rot = torch.einsum('tbef,eg->tbgf', x.grad, q)
    scale = float(x.grad.abs().max())
    assert float((x2.grad - g_rot).abs().max()) < 2e-4 * scale


def test_separate_vs_oracle():
    """Evaluation path (pit/evaluate.py:149-163) on the device vs numpy stft -> torch-CPU masks ->
    complex masking -> numpy istft."""
    from oracle import features_np, stft_np, torch_ref
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    rng = np.random.RandomState(11)
    lens = [5000, 4100, 2345]
    ys = [features_np.synthetic_mixture(rng, n)[1] for n in lens]
    torch.manual_seed(1)
    kw = dict(F=257, recurrent_layers=2, units=16, K=2)
    model = PermutationInvariantTrainingModel(**kw)
    ref = torch_ref.PITModelRef(**kw)
    ref.load_state_dict(model.state_dict())
    model = model.to(DEV).eval()
    out = model.separate([torch.from_numpy(y).to(DEV) for y in ys])
    Ys = [stft_np.stft(y, 512, 128) for y in ys]
    with torch.no_grad():
        masks = ref(dict(Y_abs=[torch.from_numpy(np.abs(Y).astype(np.float32)) for Y in Ys]))
    for b, (Y, m, n) in enumerate(zip(Ys, masks, lens)):
        Z = m.numpy()[:, :, :] * Y[:, None, :]
        z = stft_np.istft(np.transpose(Z, (1, 0, 2)), 512, 128)[:, :n]
        assert out[b].shape == (2, n)
        np.testing.assert_allclose(out[b].cpu().numpy(), z, atol=1e-4)


def test_smoke_entry():
    import __graft_entry__
    __graft_entry__.smoke()


@pytest.mark.parametrize('overlap', [True, False])
def test_trainer_rccl_path_world_size_1(g6, tmp_path, overlap):
    """The RCCL data-parallel code path (process group 'nccl', step-0 broadcast, the layer buckets of the flat gradient
    buffer all-reduced asynchronously behind the weight-gradient stream during the last micro-step's backward pass - or
    one all-reduce in optimizer_step) with world_size 1 on the GPU: must reproduce the single-process result."""
    import os
    import socket
    import torch.distributed as dist
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    batch = _batch(g6, ['Y_abs', 'X_abs', 'cos_phase_difference'])
    exs = [{k: [v[b] for b in idx] for k, v in batch.items()} for idx in g6['train_example_indices']]
    kw = dict(loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.), summary_trigger=(1000, 'iteration'),
              checkpoint_trigger=(1000, 'iteration'), stop_trigger=(3, 'iteration'), virtual_minibatch_size=2)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        model = _load(PermutationInvariantTrainingModel(F=9, recurrent_layers=2, units=4, K=2), g6, 'pit_sd_')
        t = pt.Trainer(model, tmp_path / 'dp', pt.optimizer.Adam(gradient_clipping=1.), overlap_allreduce=overlap, **kw)
        assert t.world_size == 1 and t.rank == 0
        issued = []
        real = dist.all_reduce
        dist.all_reduce = lambda tensor, *a, **k: (issued.append(tensor.numel()), real(tensor, *a, **k))[1]
        try:
            t.train(exs, device=DEV)
        finally:
            dist.all_reduce = real
        nflat = sum(p.numel() for p in model.parameters())
        # 3 optimizer steps: per step the four layer buckets (last first), or the whole buffer once
        per_step = [n for n in issued if n > 1][:4 if overlap else 1]
        assert sum(per_step) == nflat and len([n for n in issued if n > 1]) == (12 if overlap else 3), issued
        # force the collective path once (world_size 1: all_reduce(SUM) is the identity)
        dist.all_reduce(t._flat.flat, op=dist.ReduceOp.SUM)
        t._broadcast_parameters()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g6['pit_sd3_' + k], atol=2e-5, err_msg=k)


@pytest.mark.gpu
def test_device_prefetcher_overlaps_and_preserves_batches():
    """``data.DevicePrefetcher`` on the GPU: pinned host batches arrive as device tensors with the right contents while the
    consumer runs kernels in between (the copy of batch i + 1 is issued before batch i is consumed)."""
    import torch
    from padertorch_amd.data import DevicePrefetcher
    host = [dict(y=torch.full((4, 1000), float(i)).pin_memory(), tag=i) for i in range(6)]
    seen = []
    for ex in DevicePrefetcher(host, 'cuda:0'):
        assert ex['y'].is_cuda
        z = ex['y'] * 2 + 1                       # consumer work on the current stream
        seen.append((ex['tag'], float(z.min()), float(z.max())))
    assert seen == [(i, 2.0 * i + 1, 2.0 * i + 1) for i in range(6)]


@pytest.mark.gpu
def test_config4_size_step_on_the_rccl_path_next_to_a_cu_occupying_kernel(tmp_path):
    """Pre-flight of BASELINE configs[3] for the first real 8-GPU run, as far as ONE GPU can take it: the full
    model (3 x BLSTM-600) at the full per-GPU batch (64 x 4 s @ 16 kHz), 4 micro-steps per optimizer step, process group 'nccl'
    (world size 1), layer buckets all-reduced under the last micro-step's backward pass - while a third queue keeps 256 workgroups
    resident in 1.5 ms pieces (the stand-in for RCCL's channel kernels, which hold CUs beside the persistent recurrence launches).
    Two optimizer steps: no recurrence watchdog timeout, finite losses, every bucket reduced once per step, parameters moved."""
    import os
    import socket
    import torch.distributed as dist
    import padertorch_amd as pt
    from padertorch_amd import _lib
    from padertorch_amd.ops import lstm as L
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    lib = _lib.load()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        torch.manual_seed(4)
        model = PermutationInvariantTrainingModel()
        t = pt.Trainer(model, tmp_path / 'c4', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.),
                       virtual_minibatch_size=4, overlap_allreduce=True, deferred_checks=True)
        t.to(torch.device(DEV))
        t._flat = t.optimizer.use_flat_grads()
        hooks = t.enable_bucketed_allreduce()
        buckets = t._buckets
        assert buckets is not None and len(buckets.buckets) >= 4
        model.train()
        t.op_context.defer_wgrad = True          # what Trainer.train() sets
        L.warm_side_stream(torch.device(DEV))
        g = torch.Generator().manual_seed(0)
        y = (0.1 * torch.randn(64, 64000, generator=g)).to(DEV)
        s_ = (0.1 * torch.randn(64, 2, 64000, generator=g)).to(DEV)
        before = model.linear2.weight.detach().clone()
        issued = []
        real = dist.all_reduce
        dist.all_reduce = lambda tensor, *a, **k: (issued.append(tensor.numel()), real(tensor, *a, **k))[1]
        occupy = torch.cuda.Stream()
        losses = []
        try:
            for step in range(2):
                with torch.cuda.stream(occupy):
                    for _ in range(80):          # ~120 ms of occupancy: covers the optimizer step's four micro-steps
                        _lib.check(_lib.test_hooks().ptmi_test_occupy(256, 256, 0, 150000, _lib.stream(torch.device(DEV))), 'occupy')
                for m in range(4):
                    buckets.active = m == 3
                    feats = pt.ops.pit_features(y, s_)
                    loss, _, _, _ = t.train_step(model, feats, torch.device(DEV))
                    loss.backward()
                    losses.append(loss.detach())
                t.optimizer_step()
            t._check_pending(flush=True)             # raises on a timed-out recurrence launch or a non-finite step
            torch.cuda.synchronize()
        finally:
            dist.all_reduce = real
            for h in hooks:
                h.remove()
        L.check_errors()
        assert L.DEFER_WGRAD is False and L.GRAD_READY_HOOK is None and L.GRAD_USE_HOOK is None      # nothing process-global was touched
        assert all(bool(torch.isfinite(v)) for v in losses)
        nflat = t._flat.flat.numel()
        big = [n for n in issued if n > 8]
        assert sum(big) == 2 * nflat and len(big) == 2 * len(buckets.buckets), (big, nflat)
        assert float((model.linear2.weight.detach() - before).abs().max()) > 0.
    finally:
        dist.destroy_process_group()


def test_features_one_batch_ahead_on_the_prefetch_stream():
    """``DevicePrefetcher(..., to_device=<transfer + feature kernel>, release='mark')``: the features of batch n + 1 are made on the
    prefetch stream while batch n trains; the planes batch n's first projection reads are still its own, and three optimizer steps give
    bit for bit the parameters of the serial loop."""
    import padertorch_amd as pt
    from padertorch_amd.data import DevicePrefetcher
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    host = [dict(y=(0.1 * torch.randn(16, 16000, generator=g)).pin_memory(), s=(0.1 * torch.randn(16, 2, 16000, generator=g)).pin_memory())
            for _ in range(3)]
    results = []
    for ahead in (False, True):
        torch.manual_seed(1)
        model = PermutationInvariantTrainingModel(recurrent_layers=2, units=600)
        trainer = pt.Trainer(model, f'/tmp/ptmi_ahead_{int(ahead)}', pt.optimizer.Adam(gradient_clipping=1.),
                             loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.))
        trainer.to(dev)
        trainer.optimizer.use_flat_grads()

        def make(ex, d):
            return pt.ops.pit_features(ex['y'].to(d, non_blocking=True), ex['s'].to(d, non_blocking=True))
        batches = DevicePrefetcher(host, dev, to_device=make, release='mark') if ahead else (make(ex, dev) for ex in host)
        losses = []
        for feats in batches:
            assert feats['Y_abs'].packed_log1p.planes() is not None          # not retired by the batch made ahead
            loss, _, _, _ = trainer.train_step(model, feats, dev)
            loss.backward()
            trainer.optimizer_step()
            losses.append(loss.detach())
        torch.cuda.synchronize()
        pt.ops.lstm.check_errors()
        results.append((torch.stack(losses).cpu(), [p.detach().cpu().clone() for p in model.parameters()]))
    assert torch.equal(results[0][0], results[1][0])
    for a, b in zip(results[0][1], results[1][1]):
        assert torch.equal(a, b)
