#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/ by importing the REAL reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference (pure Python) is imported from /root/reference with the stand-ins of
tests/golden/ref_shim/ for its absent third-party deps (README there).  The outputs are data only
(inputs + expected outputs); neither the reference nor the shim travels to the GPU box.

Fixtures:
  g1_known_answers.json   literal answers held by the reference's own tests/doctests
  g2_stft.npz             pt.ops.STFT option grid: forward + inverse + frame/sample helpers
  g3_features.npz         pit/data.py pre_batch_transform on a seeded 2-speaker mixture
  g4_pit.npz              pit_loss / compute_pairwise_losses / pit_loss_from_loss_matrix
  g5_dc.npz               deep_clustering_loss
  g6_models.npz           tiny PIT + DC model: state_dict, inputs, outputs, losses, grads,
                          3 reference-Trainer optimizer steps (virtual_minibatch_size=2)
  g7_td_losses.npz        ops/losses/regression.py: doctest answers, seeded (K, T) signals -> every loss
                          (options grid), gradients w.r.t. the estimate, pit_loss over them, the
                          TasNet loss of a ragged batch; StftEncoder / IstftDecoder outputs
  g9_norm.npz             modules/normalization.py: normalize() outputs + reference-autograd gradients over
                          a grid of formats / axes / shift / scale / lengths; Normalization and
                          InputNormalization training steps + eval; SimpleMaskEstimator forward / loss / grads
  g11_mask_estimator.npz  contrib/jensheit/mask_estimator_example/modul.py MaskEstimator built through the reference's own
                          Configurable defaults (finalize_dogmatic_config): state_dict, ragged multi-channel inputs, every output
                          key, gradients of all parameters; with and without the VAD head / the normalisation
  g10_summary_data.npz    summary/tbx_utils.py mask_to_image / stft_to_image / spectrogram_to_image over batch_first x colour x
                          origin; data/batch.py Sorter and data/utils.py collate_fn results
  g8_logmel.npz           contrib/je/modules/features.py MelTransform (forward / inverse / maxima) and
                          the extractor front-end (stacked pt.ops.STFT -> power -> MelTransform); the
                          filterbank comes from the shim's restatement of paderbox.get_fbanks (parity
                          of that matrix to paderbox is UNPINNED; everything after it is reference code)
"""
import itertools
import json
import os
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path[:0] = [str(HERE / 'ref_shim'), str(REPO), '/root/reference']

import numpy as np  # noqa: E402
import torch  # noqa: E402

import padertorch as pt  # noqa: E402  (the reference)
from padertorch.ops.losses.source_separation import (  # noqa: E402
    compute_pairwise_losses, pit_loss_from_loss_matrix)
from padertorch.contrib.examples.source_separation.pit.model import (  # noqa: E402
    PermutationInvariantTrainingModel)
from padertorch.contrib.examples.source_separation.pit.data import pre_batch_transform  # noqa: E402
from padertorch.contrib.tcl.dc import DeepClusteringModel  # noqa: E402

torch.set_num_threads(1)
torch.use_deterministic_algorithms(True)


def g1():
    """Literal known answers copied as DATA from the reference's tests/doctests."""
    d = {
        # padertorch/contrib/cb/transform.py:219-232 (hann, size 4, shift 2, arange(8), fading full)
        'cb_stft': {
            'kwargs': dict(size=4, shift=2, window='hann', fading='full'),
            'input': list(range(8)),
            'real': [[0.5, 0., -0.5], [4., -2., 0.], [8., -4., 0.], [12., -6., 0.], [3.5, 0., -3.5]],
            'imag': [[0., 0.5, 0.], [0., 1., 0.], [0., 1., 0.], [0., 1., 0.], [0., -3.5, 0.]],
        },
        # tests/test_ops/test_stft.py:44-70 and :139-165 (samples -> frames)
        'frame_counts': [
            dict(size=1024, shift=256, window_length=1024, fading=False, samples=[1023, 1024, 1025], frames=[1, 1, 2]),
            dict(size=1024, shift=256, window_length=1024, fading=True, samples=[1023, 1024, 1025], frames=[7, 7, 8]),
            dict(size=512, shift=20, window_length=40, fading=False, samples=[1019, 1020, 1021], frames=[50, 50, 51]),
            dict(size=512, shift=20, window_length=40, fading=True, samples=[1019, 1020, 1021], frames=[52, 52, 53]),
        ],
        # padertorch/ops/_stft.py:113-114,124-125,194-195 doctest shapes
        'doctest_shapes': [
            dict(size=512, shift=20, window_length=40, rep='concat', inp=[2, 6, 203], out=[2, 6, 12, 514]),
            dict(size=512, shift=20, window_length=40, rep='complex', inp=[2, 6, 203], out=[2, 6, 12, 257]),
            dict(size=512, shift=20, window_length=40, rep='concat', inverse_inp=[2, 4, 10, 514], inverse_out=[2, 4, 180]),
        ],
        # tests/test_ops/test_losses.py:137-150
        'pit_toys': [
            dict(estimate=[[[0], [2]]], target=[[[0], [2]]], loss=0.),
            dict(estimate=[[[0], [2]]], target=[[[2], [0]]], loss=0.),
            dict(estimate=[[[0], [2]]], target=[[[-1], [0]]], loss=2.5),
            dict(estimate=[[[0], [1]]], target=[[[0], [1]]], loss=0.),
        ],
        # tests/test_ops/test_losses.py:61-84,105-116
        'dc_toys': [
            dict(embedding=[[1, 0], [0, 1], [0, 1]], target=[[1, 0], [0, 1], [0, 1]], loss=0.),
            dict(embedding=[[1, 0], [0, 1], [0, 1]], target=[[1, 0], [0, 1], [1, 0]], loss=4 / 9),
        ],
        # padertorch/ops/losses/source_separation.py:63-93 doctests (mse variants)
        'pit_doctests': [
            dict(est_shape=[4, 2, 5], axis=1, loss=1.),
            dict(est_shape=[2, 5, 4], axis=0, loss=1.),
            dict(est_shape=[5], axis=0, loss=1.),
            dict(est_shape=[4, 5, 3, 100, 128], axis=-3, loss=1.),
        ],
        # source_separation.py:262-270: -score matrix, Hungarian 'sum' -> -26
        'hungarian': dict(score=[[11., 10, 0], [4, 5, 10], [6, 0, 5]], loss_sum=-26.),
    }
    # replay them against the imported reference so a typo here cannot become "truth"
    for toy in d['pit_toys']:
        got = pt.ops.losses.pit_loss(torch.tensor(toy['estimate'], dtype=torch.float32),
                                     torch.tensor(toy['target'], dtype=torch.float32), axis=-2)
        np.testing.assert_allclose(got, toy['loss'], rtol=1e-4)
    for toy in d['dc_toys']:
        got = pt.ops.losses.deep_clustering_loss(torch.tensor(toy['embedding'], dtype=torch.float64),
                                                 torch.tensor(toy['target'], dtype=torch.float64))
        np.testing.assert_allclose(got, toy['loss'], atol=1e-12)
    for fc in d['frame_counts']:
        s = pt.ops.STFT(fc['size'], fc['shift'], window_length=fc['window_length'],
                        fading=fc['fading'], complex_representation='concat')
        for n, fr in zip(fc['samples'], fc['frames']):
            assert s(torch.rand(n)).shape[0] == fr, (fc, n)
            assert s.samples_to_frames(n) == fr, (fc, n)
    got = pit_loss_from_loss_matrix(-torch.tensor(d['hungarian']['score']), reduction='sum')
    assert float(got) == d['hungarian']['loss_sum']
    (HERE / 'g1_known_answers.json').write_text(json.dumps(d, indent=1))


STFT_GRID = [
    # size, shift, window_length, window
    (512, 128, 512, 'blackman'),
    (1024, 256, 1024, 'blackman'),
    (512, 20, 40, 'hamming'),
    (256, 10, 20, 'hann'),
    (64, 24, 50, 'hann'),          # window_length % shift != 0
    (128, 48, 100, 'blackman'),
]


def g2():
    rng = np.random.RandomState(2)
    out = {}
    cases = []
    for (size, shift, wl, win) in STFT_GRID:
        # long enough for fading=False/pad=False (the reference's conv1d needs T >= window_length)
        shape = (2, 400) if size <= 256 else (1, wl + 3 * shift + 5)
        out[f'x_s{size}_h{shift}'] = rng.standard_normal(shape)
    for (size, shift, wl, win), fading, pad in itertools.product(
            STFT_GRID, ['full', 'half', False], [True, False]):
        name = f's{size}_h{shift}_l{wl}_{win}_{fading}_{int(pad)}'
        x = out[f'x_s{size}_h{shift}']
        s = pt.ops.STFT(size, shift, window=win, window_length=wl, fading=fading, pad=pad,
                        complex_representation='complex')
        X = s(torch.from_numpy(x))
        xi = s.inverse(X)
        out[name + '_X'] = X.numpy()
        out[name + '_xi'] = xi.numpy()
        cases.append(dict(name=name, x=f'x_s{size}_h{shift}', size=size, shift=shift, window_length=wl, window=win,
                          fading=fading, pad=pad,
                          frames=int(s.samples_to_frames(x.shape[-1])),
                          samples_back=int(s.frames_to_samples(X.shape[-2]))))
        assert cases[-1]['frames'] == X.shape[-2]
        assert cases[-1]['samples_back'] == xi.shape[-1]
    # the other two representations once (concat, stacked), fp32 input like the doctests
    x32 = torch.from_numpy(out['x_s512_h128'].astype(np.float32))
    for rep in ['concat', 'stacked']:
        s = pt.ops.STFT(512, 128, complex_representation=rep)
        X = s(x32)
        out[f'rep_{rep}_X'] = X.numpy()
        out[f'rep_{rep}_xi'] = s.inverse(X).numpy()
    out['cases'] = np.array(json.dumps(cases))
    np.savez(HERE / 'g2_stft.npz', **out)


def g3():
    rng = np.random.RandomState(3)
    s = (0.1 * rng.standard_normal((2, 1500))).astype(np.float32)
    y = s.sum(0)
    f = pre_batch_transform(dict(audio_data=dict(speech_source=s, observation=y), example_id='g3'))
    np.savez(HERE / 'g3_features.npz', s=s, y=y, Y=f['Y'], X_abs=f['X_abs'], Y_abs=f['Y_abs'],
             cos_phase_difference=f['cos_phase_difference'], num_frames=f['num_frames'])


def g4():
    rng = np.random.RandomState(4)
    out = {}
    names = []
    for name, shape, axis in [('a', (7, 2, 5), -2), ('b', (5, 3, 4), -2), ('c', (3, 6, 4), 0),
                              ('d', (6, 4, 3), 1)]:
        est = rng.standard_normal(shape).astype(np.float32)
        tgt = rng.standard_normal(shape).astype(np.float32)
        loss, perm = pt.ops.losses.pit_loss(torch.from_numpy(est), torch.from_numpy(tgt), axis=axis,
                                            return_permutation=True)
        pw = compute_pairwise_losses(torch.from_numpy(est), torch.from_numpy(tgt), axis=axis)
        hl, col = pit_loss_from_loss_matrix(pw, return_permutation=True)
        out.update({f'{name}_est': est, f'{name}_tgt': tgt, f'{name}_loss': loss.numpy(),
                    f'{name}_perm': np.array(perm), f'{name}_axis': axis,
                    f'{name}_pairwise': pw.numpy(), f'{name}_hungarian_loss': hl.numpy(),
                    f'{name}_hungarian_col': np.asarray(col)})
        names.append(name)
    # tie: identical estimates -> every permutation has the same loss -> first (identity) wins
    est = np.repeat(rng.standard_normal((4, 1, 3)).astype(np.float32), 3, axis=1)
    tgt = rng.standard_normal((4, 3, 3)).astype(np.float32)
    loss, perm = pt.ops.losses.pit_loss(torch.from_numpy(est), torch.from_numpy(tgt), axis=1,
                                        return_permutation=True)
    out.update(tie_est=est, tie_tgt=tgt, tie_loss=loss.numpy(), tie_perm=np.array(perm), tie_axis=1)
    names.append('tie')
    # gradient of the PIT loss wrt the estimate (autograd of the reference)
    est = torch.from_numpy(out['b_est']).clone().requires_grad_(True)
    pt.ops.losses.pit_loss(est, torch.from_numpy(out['b_tgt']), axis=-2).backward()
    out['b_grad'] = est.grad.numpy()
    out['names'] = np.array(json.dumps(names))
    np.savez(HERE / 'g4_pit.npz', **out)


def g5():
    rng = np.random.RandomState(5)
    x = rng.standard_normal((100, 20)).astype(np.float32)
    x /= np.linalg.norm(x, axis=-1, keepdims=True)
    t = np.eye(3, dtype=np.float32)[rng.randint(0, 3, size=100)]
    xt = torch.from_numpy(x).clone().requires_grad_(True)
    loss = pt.ops.losses.deep_clustering_loss(xt, torch.from_numpy(t))
    loss.backward()
    loss64 = pt.ops.losses.deep_clustering_loss(torch.from_numpy(x).double(), torch.from_numpy(t).double())
    np.savez(HERE / 'g5_dc.npz', x=x, t=t, loss=loss.detach().numpy(), loss64=loss64.numpy(),
             grad=xt.grad.numpy())


def _sd_to_np(sd, prefix):
    return {prefix + k: v.detach().numpy().copy() for k, v in sd.items()}


def g6():
    out = {}
    rng = np.random.RandomState(6)
    Ts = [6, 5, 3]
    F, K, E = 9, 2, 3
    batch = dict(
        Y_abs=[np.abs(rng.standard_normal((T, F))).astype(np.float32) for T in Ts],
        X_abs=[np.abs(rng.standard_normal((T, K, F))).astype(np.float32) for T in Ts],
        cos_phase_difference=[np.cos(rng.uniform(-3, 3, (T, K, F))).astype(np.float32) for T in Ts],
        target_mask=[np.eye(K, dtype=np.float32)[rng.randint(0, K, (T, F))].transpose(0, 2, 1).copy()
                     for T in Ts],
        num_frames=Ts,
    )
    for k in ['Y_abs', 'X_abs', 'cos_phase_difference', 'target_mask']:
        for b, a in enumerate(batch[k]):
            out[f'in_{k}_{b}'] = a
    out['Ts'] = np.array(Ts)

    # ---- PIT model
    torch.manual_seed(60)
    model = PermutationInvariantTrainingModel(F=F, recurrent_layers=2, units=4, K=K)
    out.update(_sd_to_np(model.state_dict(), 'pit_sd_'))
    ex = model.example_to_device(batch, 'cpu')
    masks = model(ex)
    review = model.review(ex, masks)
    for b, m in enumerate(masks):
        out[f'pit_mask_{b}'] = m.detach().numpy()
    out['pit_mse_loss'] = review['losses']['pit_mse_loss'].detach().numpy()
    out['pit_ips_loss'] = review['losses']['pit_ips_loss'].detach().numpy()
    # per-example single losses: "batch loss == mean of single-example losses" (test_bss.py:153-192)
    singles = []
    for b in range(len(Ts)):
        exb = {k: [v[b]] for k, v in ex.items()}
        rb = model.review(exb, model(exb))
        singles.append([float(rb['losses']['pit_mse_loss']), float(rb['losses']['pit_ips_loss'])])
    out['pit_single_losses'] = np.array(singles)
    (1.0 * review['losses']['pit_ips_loss']).backward()
    for n, p in model.named_parameters():
        out[f'pit_grad_{n}'] = p.grad.numpy().copy()

    # ---- 3 optimizer steps under the reference Trainer (virtual_minibatch_size=2, clip=1)
    torch.manual_seed(60)
    model = PermutationInvariantTrainingModel(F=F, recurrent_layers=2, units=4, K=K)
    with tempfile.TemporaryDirectory() as tmp:
        trainer = pt.Trainer(model, tmp, pt.optimizer.Adam(gradient_clipping=1.),
                             loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.),
                             summary_trigger=(1000, 'iteration'), checkpoint_trigger=(1000, 'iteration'),
                             stop_trigger=(3, 'iteration'), virtual_minibatch_size=2)
        # 6 examples = 3 optimizer steps x 2 accumulated micro-steps; examples = sub-batches
        exs = [{k: [v[b] for b in idx] for k, v in batch.items()}
               for idx in [(0, 1, 2), (0, 1), (1, 2), (0, 2), (0,), (1,)]]
        trainer.train(exs, device='cpu')
        assert trainer.iteration == 3, trainer.iteration
    out.update(_sd_to_np(model.state_dict(), 'pit_sd3_'))
    out['train_example_indices'] = np.array(json.dumps([(0, 1, 2), (0, 1), (1, 2), (0, 2), (0,), (1,)]))

    # ---- DC model
    torch.manual_seed(61)
    dc = DeepClusteringModel(F=F, recurrent_layers=1, units=4, E=E)
    out.update(_sd_to_np(dc.state_dict(), 'dc_sd_'))
    ex = dc.example_to_device(batch, 'cpu')
    emb = dc(ex)
    review = dc.review(ex, emb)
    for b, m in enumerate(emb):
        out[f'dc_emb_{b}'] = m.detach().numpy()
    out['dc_loss'] = review['losses']['dc_loss'].detach().numpy()
    review['losses']['dc_loss'].backward()
    for n, p in dc.named_parameters():
        out[f'dc_grad_{n}'] = p.grad.numpy().copy()
    np.savez(HERE / 'g6_models.npz', **out)


def g7():
    """Time-domain regression losses + their PIT use + the TasNet coders."""
    import functools
    from padertorch.ops.losses import regression as R
    from padertorch.contrib.examples.source_separation.tasnet.tas_coders import StftEncoder, IstftDecoder
    rng = np.random.RandomState(7)
    out = {}
    fns = {
        'mse': R.mse_loss, 'log-mse': R.log_mse_loss, 'log1p-mse': R.log1p_mse_loss, 'sdr': R.sdr_loss,
        'si-sdr': R.si_sdr_loss, 'sa-sdr': R.source_aggregated_sdr_loss,
        'log-mse@20': functools.partial(R.log_mse_loss, soft_sdr_max=20),
        'sdr@20': functools.partial(R.sdr_loss, soft_sdr_max=20),
        'si-sdr@30': functools.partial(R.si_sdr_loss, soft_sdr_max=30),
        'si-sdr-oi': functools.partial(R.si_sdr_loss, offset_invariant=True),
        'si-sdr-gs': functools.partial(R.si_sdr_loss, grad_stop=True),
        'si-sdr-sum': functools.partial(R.si_sdr_loss, reduction='sum'),
        'log-mse-mean': functools.partial(R.log_mse_loss, reduction='mean'),
    }
    # doctest pair (regression.py:60-67, 119-124, 148-153, 207-212, 333-338, 356-361)
    de = torch.tensor([[1., 2, 3], [4, 5, 6]])
    dt = torch.tensor([[2., 3, 4], [4, 0, 6]])
    out['doc_estimate'], out['doc_target'] = de.numpy(), dt.numpy()
    for n, f in fns.items():
        out[f'doc/{n}'] = f(de, dt).numpy()
    for n in ('mse', 'log-mse', 'log1p-mse', 'sdr', 'si-sdr'):
        out[f'doc_none/{n}'] = fns[n](de, dt, reduction=None).numpy()
    # seeded signals: K sources of T samples, estimate = mixture of the targets + noise
    cases = []
    for K, T in ((2, 777), (3, 400), (4, 257)):
        tgt = (0.3 * rng.randn(K, T) + 0.05).astype(np.float32)
        mix = np.eye(K)[::-1] * 0.8 + 0.15 * rng.rand(K, K)          # permuted, leaky
        est = (mix @ tgt + 0.05 * rng.randn(K, T)).astype(np.float32)
        key = f'K{K}_T{T}'
        cases.append(key)
        out[f'{key}/estimate'], out[f'{key}/target'] = est, tgt
        for n, f in fns.items():
            e = torch.tensor(est, requires_grad=True)
            t = torch.tensor(tgt, requires_grad=True)
            loss = f(e, t)
            loss.backward()
            out[f'{key}/{n}/loss'] = loss.detach().numpy()
            out[f'{key}/{n}/grad_estimate'] = e.grad.numpy()
            if n in ('si-sdr', 'mse'):
                out[f'{key}/{n}/grad_target'] = t.grad.numpy()
            e64 = torch.tensor(est, dtype=torch.float64)
            out[f'{key}/{n}/loss64'] = f(e64, torch.tensor(tgt, dtype=torch.float64)).numpy()
            if True:
                e2 = torch.tensor(est, requires_grad=True)
                pl, perm = pt.ops.losses.pit_loss(e2, torch.tensor(tgt), axis=0, loss_fn=f, return_permutation=True)
                pl.backward()
                out[f'{key}/{n}/pit_loss'] = pl.detach().numpy()
                out[f'{key}/{n}/pit_perm'] = np.array(perm)
                if n in ('si-sdr', 'log-mse', 'sa-sdr'):
                    out[f'{key}/{n}/pit_grad_estimate'] = e2.grad.numpy()
    out['cases'] = np.array(json.dumps(cases))
    out['names'] = np.array(json.dumps(list(fns)))
    # TasNet.loss on a ragged batch (tasnet/model.py:154-176), B=3, K=2, padded to 1200
    B, K, T = 3, 2, 1200
    num_samples = [1200, 1111, 640]
    s = (0.3 * rng.randn(B, K, T)).astype(np.float32)
    x = (s[:, ::-1] * 0.9 + 0.1 * s + 0.05 * rng.randn(B, K, T)).astype(np.float32)
    x[1] = (s[1] * 0.9 + 0.05 * rng.randn(K, T)).astype(np.float32)       # identity permutation
    xt = torch.tensor(x, requires_grad=True)
    losses = {k: [] for k in ('si-sdr', 'log-mse', 'log1p-mse')}
    tas = {'si-sdr': R.si_sdr_loss, 'log-mse': R.log_mse_loss, 'log1p-mse': R.log1p_mse_loss}
    for n_, est_, tgt_ in zip(num_samples, xt, torch.tensor(s)):
        for k, f in tas.items():
            losses[k].append(pt.ops.losses.pit_loss(est_[..., :n_], tgt_[..., :n_], axis=0, loss_fn=f))
    tl = {k: torch.mean(torch.stack(v)) for k, v in losses.items()}
    (tl['si-sdr'] + 0.5 * tl['log-mse'] + 0.25 * tl['log1p-mse']).backward()
    out['tas/x'], out['tas/s'], out['tas/num_samples'] = x, s, np.array(num_samples)
    for k, v in tl.items():
        out[f'tas/{k}'] = v.detach().numpy()
    out['tas/grad_x'] = xt.grad.numpy()          # of si-sdr + 0.5 log-mse + 0.25 log1p-mse
    # StftEncoder / IstftDecoder (tasnet/tas_coders.py:138-240)
    mixture = torch.tensor(rng.rand(2, 3, 203).astype(np.float32))
    enc = StftEncoder(feature_size=258)
    encoded, num_frames = enc(mixture, [203, 150])
    out['coder/mixture'] = mixture.numpy()
    out['coder/encoded'] = encoded.numpy()
    out['coder/num_frames'] = num_frames.numpy()
    stft_signal = torch.tensor(rng.rand(2, 4, 258, 10).astype(np.float32))
    out['coder/stft_signal'] = stft_signal.numpy()
    out['coder/decoded'] = IstftDecoder(feature_size=258)(stft_signal).numpy()
    enc2 = StftEncoder(window_length=16, feature_size=66, stride=4)
    out['coder/encoded_16_66_4'] = enc2(mixture).numpy()
    out['coder/roundtrip_16_66_4'] = IstftDecoder(window_length=16, feature_size=66, stride=4)(enc2(mixture)).numpy()
    # complex signals through the scale-invariant loss (regression.py:21-24,178-296: the scaling factor is the UNCONJUGATED
    # product sum(e t) / sum |t|^2, a complex number): values and the reference-autograd gradient w.r.t. the estimate
    rng_c = np.random.RandomState(77)         # (its own stream: everything above keeps its values)
    ec = (rng_c.randn(3, 2, 333) + 1j * rng_c.randn(3, 2, 333)).astype(np.complex64)
    tc = (rng_c.randn(3, 2, 333) + 1j * rng_c.randn(3, 2, 333)).astype(np.complex64)
    tc = (tc + 0.6 * ec).astype(np.complex64)
    out['complex/estimate'], out['complex/target'] = ec, tc
    cnames = []
    for name, kw in (('plain', {}), ('offset', dict(offset_invariant=True)), ('grad_stop', dict(grad_stop=True)),
                     ('soft30', dict(soft_sdr_max=30)), ('all', dict(offset_invariant=True, grad_stop=True, soft_sdr_max=20, reduction='sum')),
                     ('none', dict(reduction=None))):
        e_ = torch.tensor(ec, requires_grad=True)
        val = R.si_sdr_loss(e_, torch.tensor(tc), **kw)
        val.sum().backward()
        out[f'complex/si_sdr/{name}/value'] = val.detach().numpy()
        out[f'complex/si_sdr/{name}/grad'] = e_.grad.numpy()
        cnames.append(dict(name=name, kwargs=kw))
    out['complex/cases'] = np.array(json.dumps(cnames))
    np.savez_compressed(HERE / 'g7_td_losses.npz', **out)


def g8():
    from padertorch.contrib.je.modules.features import MelTransform
    rng = np.random.RandomState(8)
    out = {}
    cfgs = []
    for sr, size, nf, lo, hi, htk, log in ((16000, 512, 40, 50., None, True, True), (8000, 512, 80, 0., -200., True, True),
                                           (16000, 1024, 64, 50., 7600., False, True), (16000, 256, 20, 50., None, True, False)):
        key = f'sr{sr}_n{size}_m{nf}'
        cfgs.append(dict(key=key, sample_rate=sr, stft_size=size, number_of_filters=nf, lowest_frequency=lo,
                         highest_frequency=hi, htk_mel=htk, log=log))
        mt = MelTransform(sr, size, nf, lowest_frequency=lo, highest_frequency=hi, htk_mel=htk, log=log)
        mt.eval()
        spec = torch.tensor((rng.rand(2, 3, 11, size // 2 + 1) ** 2).astype(np.float32))
        y, maxima = mt(spec, return_maxima=True)
        out[f'{key}/fbanks'] = mt.fbanks.detach().numpy()
        out[f'{key}/spec'] = spec.numpy()
        out[f'{key}/mel'] = y.numpy()
        out[f'{key}/maxima'] = maxima.numpy()
        out[f'{key}/inverse'] = mt.inverse(y).numpy()
    out['configs'] = np.array(json.dumps(cfgs))
    # extractor front-end (features.py:171-176) on waveforms: stacked STFT -> sum of squares -> mel -> log
    x = (0.1 * rng.randn(3, 4000)).astype(np.float32)
    stft = pt.ops.STFT(512, 128, window_length=512, complex_representation='stacked')
    mt = MelTransform(16000, 512, 80)
    mt.eval()
    X = stft(torch.tensor(x))
    out['front/x'] = x
    out['front/logmel'] = mt(torch.sum(X ** 2, dim=(-1,))).numpy()
    mt1 = MelTransform(16000, 512, 40, log=False)
    mt1.eval()
    out['front/mel_magnitude'] = mt1(torch.sum(X ** 2, dim=(-1,)).sqrt()).numpy()
    stft2 = pt.ops.STFT(1024, 256, window_length=800, window='hann', fading='half', complex_representation='stacked')
    mt2 = MelTransform(16000, 1024, 64)
    mt2.eval()
    out['front/logmel_1024_256_800'] = mt2(torch.sum(stft2(torch.tensor(x)) ** 2, dim=(-1,))).numpy()
    np.savez_compressed(HERE / 'g8_logmel.npz', **out)


def g9():
    from padertorch.modules.normalization import normalize, Normalization, InputNormalization
    from padertorch.contrib.examples.speech_enhancement.mask_estimator.model import SimpleMaskEstimator
    rng = np.random.RandomState(9)
    out, cases = {}, []
    grid = [
        # data_format, shape, statistics axes, independent axes, lengths
        ('bct', (2, 3, 5), 'bt', 'c', [5, 3]),              # the reference's own test (test_norm.py:38-71)
        ('btf', (3, 7, 6), 't', 'f', [7, 4, 2]),            # SimpleMaskEstimator's normalisation
        ('bcft', (2, 3, 4, 9), 'bft', 'c', [9, 6]),
        ('bcft', (2, 3, 4, 9), 'bt', 'cf', [9, 5]),
        ('bft', (4, 5, 8), 'f', None, None),
        ('tbf', (6, 3, 5), 'tb', 'f', [6, 2, 1]),
        ('bcft', (2, 2, 3, 70), 'bf', 'c', [70, 33]),       # alternating kept / reduced axes
    ]
    for fmt, shape, stat, indep, lens in grid:
        for shift, scale in ((True, True), (True, False), (False, True), (False, False)):
            key = f'{fmt}_{stat}_{indep}_{int(shift)}{int(scale)}'
            x = torch.tensor(rng.randn(*shape).astype(np.float32) * 1.5 + 0.7, requires_grad=True)
            gshape = [shape[i] if (indep and fmt[i] in indep) else 1 for i in range(len(shape))]
            gamma = torch.tensor((1 + 0.3 * rng.randn(*gshape)).astype(np.float32), requires_grad=True) if indep and scale else None
            beta = torch.tensor((0.3 * rng.randn(*gshape)).astype(np.float32), requires_grad=True) if indep and shift else None
            axes = [fmt.index(a) for a in stat]
            b_ax = fmt.index('b')
            t_ax = fmt.index('t')
            y, mean, power, n = normalize(x, gamma, beta, axes, b_ax, t_ax, lens, shift, scale, 1e-3)
            w = torch.tensor(rng.randn(*shape).astype(np.float32))
            (y * w).sum().backward()
            cases.append(dict(key=key, fmt=fmt, shape=list(shape), stat=stat, indep=indep, lens=lens, shift=shift,
                              scale=scale, axes=axes, b_ax=b_ax, t_ax=t_ax))
            out[f'{key}/x'], out[f'{key}/w'] = x.detach().numpy(), w.numpy()
            out[f'{key}/y'], out[f'{key}/mean'] = y.detach().numpy(), mean.detach().numpy()
            out[f'{key}/power'], out[f'{key}/n'] = power.detach().numpy(), n.detach().numpy()
            out[f'{key}/grad_x'] = x.grad.numpy()
            if gamma is not None:
                out[f'{key}/gamma'], out[f'{key}/grad_gamma'] = gamma.detach().numpy(), gamma.grad.numpy()
            if beta is not None:
                out[f'{key}/beta'], out[f'{key}/grad_beta'] = beta.detach().numpy(), beta.grad.numpy()
    out['cases'] = np.array(json.dumps(cases))
    # module behaviour: three training steps (running statistics), then eval, for both classes and momenta
    mods = []
    for cls, name in ((Normalization, 'norm'), (InputNormalization, 'inorm')):
        for momentum in (0.5, None):
            key = f'{name}_m{momentum}'
            mods.append(dict(key=key, cls=name, momentum=momentum))
            m = cls(data_format='bct', shape=(None, 4, None), statistics_axis='bt', momentum=momentum)
            m.train()
            with torch.no_grad():
                m.gamma.copy_(torch.tensor((1 + 0.2 * rng.randn(1, 4, 1)).astype(np.float32)))
                m.beta.copy_(torch.tensor((0.2 * rng.randn(1, 4, 1)).astype(np.float32)))
            out[f'{key}/gamma'], out[f'{key}/beta'] = m.gamma.detach().numpy(), m.beta.detach().numpy()
            for step, lens in enumerate(([6, 4, 1], [6, 6, 6], [3, 2, 2])):
                x = torch.tensor((rng.randn(3, 4, 6) * (1 + step) + step).astype(np.float32), requires_grad=True)
                y = m(x, lens)
                y.sum().backward()
                out[f'{key}/s{step}/x'], out[f'{key}/s{step}/lens'] = x.detach().numpy(), np.array(lens)
                out[f'{key}/s{step}/y'], out[f'{key}/s{step}/grad_x'] = y.detach().numpy(), x.grad.numpy()
                for b in ('num_tracked_values', 'running_mean', 'running_power'):
                    out[f'{key}/s{step}/{b}'] = getattr(m, b).detach().numpy().copy()
            m.eval()
            x = torch.tensor(rng.randn(2, 4, 5).astype(np.float32), requires_grad=True)
            y = m(x, [5, 2])
            (y ** 2).sum().backward()
            out[f'{key}/eval/x'], out[f'{key}/eval/y'] = x.detach().numpy(), y.detach().numpy()
            out[f'{key}/eval/grad_x'] = x.grad.numpy()
            out[f'{key}/eval/grad_gamma'] = m.gamma.grad.numpy()
            out[f'{key}/eval/inverse'] = m.inverse(y.detach(), [5, 2]).detach().numpy()
    out['modules'] = np.array(json.dumps(mods))
    # SimpleMaskEstimator (speech_enhancement/mask_estimator/model.py): eval-mode forward + loss + grads
    torch.manual_seed(9)
    me = SimpleMaskEstimator(17, num_units=32, dropout=0.)
    me.eval()
    for k, v in me.state_dict().items():
        out[f'me/sd/{k}'] = v.numpy()
    obs = torch.tensor(np.abs(rng.randn(3, 11, 17)).astype(np.float32))
    batch = dict(observation_abs=obs, speech_mask_target=torch.tensor(rng.rand(3, 11, 17).astype(np.float32)),
                 noise_mask_target=torch.tensor(rng.rand(3, 11, 17).astype(np.float32)))
    o = me(batch)
    loss = me.review(batch, o)['loss']
    loss.backward()
    out['me/observation_abs'] = obs.numpy()
    out['me/speech_mask_target'], out['me/noise_mask_target'] = batch['speech_mask_target'].numpy(), batch['noise_mask_target'].numpy()
    out['me/speech_mask_prediction'] = o['speech_mask_prediction'].detach().numpy()
    out['me/noise_mask_prediction'] = o['noise_mask_prediction'].detach().numpy()
    out['me/loss'] = loss.detach().numpy()
    for k, p_ in me.named_parameters():
        out[f'me/grad/{k}'] = p_.grad.numpy()
    np.savez_compressed(HERE / 'g9_norm.npz', **out)


def g10():
    """summary/tbx_utils.py image helpers (grayscale and viridis), data/batch.py Sorter, data/utils.py collate_fn."""
    import warnings
    from padertorch.summary.tbx_utils import mask_to_image, stft_to_image, spectrogram_to_image
    from padertorch.data.batch import Sorter
    from padertorch.data.utils import collate_fn
    rng = np.random.RandomState(10)
    out, cases = {}, []
    mask = rng.rand(7, 3, 5).astype(np.float32) * 1.2 - 0.1            # some values outside [0, 1]
    spec = (rng.randn(7, 3, 5) + 1j * rng.randn(7, 3, 5)).astype(np.complex64) * np.exp(3 * rng.randn(7, 3, 5)).astype(np.float32)
    out['mask'], out['spec'] = mask, spec
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for bf in (False, True):
            for color in (None, 'viridis'):
                for origin in ('lower', 'upper'):
                    key = f'bf{int(bf)}_{color}_{origin}'
                    cases.append(dict(key=key, batch_first=bf, color=color, origin=origin))
                    m = mask.transpose(1, 0, 2) if bf else mask
                    z = spec.transpose(1, 0, 2) if bf else spec
                    out[f'{key}/mask3'] = mask_to_image(m, bf, color, origin)
                    out[f'{key}/mask2'] = mask_to_image(mask[:, 1], bf, color, origin)
                    out[f'{key}/stft3'] = stft_to_image(z, bf, color, origin)
                    out[f'{key}/stft2_60'] = stft_to_image(spec[:, 2], bf, color, origin, 60)
                    out[f'{key}/abs3'] = stft_to_image(np.abs(z), bf, color, origin)
                    out[f'{key}/pow_lin'] = spectrogram_to_image(np.abs(z) ** 2, bf, color, origin, log=False)
    out['cases'] = np.array(json.dumps(cases))
    # structural helpers: results as JSON (the doctest inputs of data/batch.py:141-143 and data/utils.py:31-40 + a ragged one)
    batch = [{'value': x, 'num_samples': n} for x, n in [(5, 10), (1, 30), (3, 20), (2, 20)]]
    structural = dict(
        sorter_value=[list(map(dict, [Sorter('value')(batch)][0]))][0],
        sorter_default=list(Sorter()(batch)),
        sorter_ascending=list(Sorter('value', reverse=False)(batch)),
        collate_flat=collate_fn([{'a': 1}, {'a': 2}]),
        collate_tuple=collate_fn(({'a': 1}, {'a': 2})),
        collate_tuple_is_tuple=isinstance(collate_fn(({'a': 1}, {'a': 2}))['a'], tuple),
        collate_nested=collate_fn([{'a': {'b': [1, 2]}}, {'a': {'b': [3, 4]}}]),
    )
    out['structural'] = np.array(json.dumps(structural))
    np.savez_compressed(HERE / 'g10_summary_data.npz', **out)


def g11():
    """contrib/jensheit MaskEstimator (modul.py:45-158): Normalization -> StatefulLSTM -> fully_connected_stack -> masks."""
    from padertorch.contrib.jensheit.mask_estimator_example.modul import MaskEstimator
    rng = np.random.RandomState(11)
    out, cases = {}, []
    F = 9
    grid = [
        dict(key='default', C=2, frames=[9, 7, 4], updates={}),
        dict(key='vad', C=1, frames=[8, 8, 5, 2], updates=dict(vad=True)),
        dict(key='nonorm_uni', C=3, frames=[6], updates=dict(normalization=None, separate_masks=False, output_activation='tanh',
                                                            recurrent=dict(bidirectional=False))),
    ]
    for case in grid:
        key, C, frames = case['key'], case['C'], case['frames']
        torch.manual_seed(11)
        upd = dict(num_features=F, recurrent=dict(hidden_size=8), fully_connected=dict(hidden_size=[16, 12, 16], dropout=0.))
        for k, v in case['updates'].items():
            if isinstance(v, dict):
                upd[k] = {**upd.get(k, {}), **v}
            else:
                upd[k] = v
        upd['fully_connected']['input_size'] = 16 if upd.get('recurrent', {}).get('bidirectional', True) else 8
        # (Configurable.get_config needs `sacred`, which this image lacks: the reference's own finalize_dogmatic_config fills
        #  the defaults, the update entries override them, the factories are called as from_config would)
        cfg = dict(num_features=F)
        MaskEstimator.finalize_dogmatic_config(cfg)
        for k, v in upd.items():
            if isinstance(v, dict) and isinstance(cfg.get(k), dict):
                cfg[k].update(v)
            else:
                cfg[k] = v

        def build(node):
            if isinstance(node, dict) and 'factory' in node:
                return node['factory'](**{k: build(v) for k, v in node.items() if k != 'factory'})
            return node
        me = MaskEstimator(**{k: build(v) for k, v in cfg.items()})
        me.eval()
        for k, v in me.state_dict().items():
            out[f'{key}/sd/{k}'] = v.numpy()
        x = [torch.tensor(np.abs(rng.randn(C, n, F)).astype(np.float32) * 2) for n in frames]
        o = me(x, frames)
        loss = 0.
        for i, (k, v) in enumerate(sorted(o.items())):
            w = torch.tensor(rng.randn(*v.shape).astype(np.float32))
            out[f'{key}/w/{k}'] = w.numpy()
            out[f'{key}/out/{k}'] = v.detach().numpy()
            loss = loss + (v * w).sum()
        loss.backward()
        for b, t in enumerate(x):
            out[f'{key}/x{b}'] = t.numpy()
        out[f'{key}/loss'] = loss.detach().numpy()
        for k, p_ in me.named_parameters():
            out[f'{key}/grad/{k}'] = p_.grad.numpy() if p_.grad is not None else np.zeros(0, np.float32)
        cases.append(dict(key=key, C=C, frames=frames, updates=case['updates'], F=F))
    out['cases'] = np.array(json.dumps(cases))
    np.savez_compressed(HERE / 'g11_mask_estimator.npz', **out)


if __name__ == '__main__':
    assert os.path.isdir('/root/reference'), 'run in the build container'
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11):
        if only and fn.__name__ not in only:
            continue
        fn()
        print('wrote', fn.__name__)
    for p in sorted(HERE.glob('g*.*')):
        print(f'{p.name:28s} {p.stat().st_size / 1024:8.1f} KiB')
