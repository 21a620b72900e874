"""HIP time-domain regression losses (ops/losses/regression.py) and the TasNet coders vs the
reference goldens g7 and the oracle (GPU, through the C ABI)."""
import functools

import numpy as np
import pytest
import torch

from oracle import losses_np as L

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def variants():
    from padertorch_amd.ops.losses import regression as R
    return {
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


def dev(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).requires_grad_(grad)


def close_grad(got, want, rel=2e-4):
    """gradients: |diff| <= rel * max|want| (fp32 reference autograd vs fp64 statistics)."""
    got, want = got.detach().cpu().numpy(), np.asarray(want)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= rel * np.abs(want).max() + 1e-12, np.abs(got - want).max() / np.abs(want).max()


def test_doctest_answers(g7):
    V = variants()
    de, dt = dev(g7['doc_estimate']), dev(g7['doc_target'])
    for n in g7['names']:
        got = V[n](de, dt)
        assert got.dtype == torch.float32 and got.dim() == 0
        np.testing.assert_allclose(got.item(), g7[f'doc/{n}'], rtol=1e-5, atol=1e-5)
    for n in ('mse', 'log-mse', 'log1p-mse', 'sdr', 'si-sdr'):
        np.testing.assert_allclose(V[n](de, dt, reduction=None).cpu().numpy(), g7[f'doc_none/{n}'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(V['sdr'](dt, dt, soft_sdr_max=20).item(), -20., atol=1e-5)
    np.testing.assert_allclose(V['si-sdr'](dt, dt, soft_sdr_max=20).item(), -20., atol=1e-5)
    with pytest.raises(ValueError):
        V['mse'](de, dt, reduction='prod')
    with pytest.raises(RuntimeError):          # no CPU fallback
        V['mse'](de.cpu(), dt.cpu())


def test_losses_and_gradients_vs_reference(g7):
    V = variants()
    for key in g7['cases']:
        for n in g7['names']:
            e, t = dev(g7[f'{key}/estimate'], True), dev(g7[f'{key}/target'], True)
            loss = V[n](e, t)
            # the reference's own fp32 run is the looser pin, its fp64 run the tight one
            np.testing.assert_allclose(loss.item(), g7[f'{key}/{n}/loss64'], rtol=2e-6, atol=2e-6)
            np.testing.assert_allclose(loss.item(), g7[f'{key}/{n}/loss'], rtol=2e-5, atol=2e-5)
            loss.backward()
            close_grad(e.grad, g7[f'{key}/{n}/grad_estimate'])
            if f'{key}/{n}/grad_target' in g7:
                close_grad(t.grad, g7[f'{key}/{n}/grad_target'])


def test_pit_loss_over_time_domain_losses(g7):
    from padertorch_amd.ops import pit_loss
    V = variants()
    for key in g7['cases']:
        for n in g7['names']:
            e = dev(g7[f'{key}/estimate'], True)
            loss, perm = pit_loss(e, dev(g7[f'{key}/target']), axis=0, loss_fn=V[n], return_permutation=True)
            assert list(perm) == list(g7[f'{key}/{n}/pit_perm']), (key, n)      # exact
            np.testing.assert_allclose(loss.item(), g7[f'{key}/{n}/pit_loss'], rtol=2e-5, atol=2e-5)
            if f'{key}/{n}/pit_grad_estimate' in g7:
                loss.backward()
                close_grad(e.grad, g7[f'{key}/{n}/pit_grad_estimate'])
    # a loss function the module does not know keeps the reference's brute-force path
    e, t = dev(g7['K3_T400/estimate']), dev(g7['K3_T400/target'])
    got, perm = pit_loss(e, t, axis=0, loss_fn=lambda a, b: V['si-sdr'](a, b) * 1.0, return_permutation=True)
    assert list(perm) == list(g7['K3_T400/si-sdr/pit_perm'])
    np.testing.assert_allclose(got.item(), g7['K3_T400/si-sdr/pit_loss'], rtol=2e-5)


def test_tasnet_loss_ragged_batch(g7):
    from padertorch_amd.contrib.examples.source_separation.tasnet import tasnet_loss
    x = dev(g7['tas/x'], True)
    out = tasnet_loss({'s': dev(g7['tas/s']), 'num_samples': g7['tas/num_samples'].tolist()}, {'out': x})
    for k in ('si-sdr', 'log-mse', 'log1p-mse'):
        np.testing.assert_allclose(out[k].item(), g7[f'tas/{k}'], rtol=2e-5, atol=2e-5)
    (out['si-sdr'] + 0.5 * out['log-mse'] + 0.25 * out['log1p-mse']).backward()
    close_grad(x.grad, g7['tas/grad_x'])
    n = g7['tas/num_samples']
    for b in range(len(n)):                                   # nothing leaks into the padding
        assert float(x.grad[b, :, n[b]:].abs().sum()) == 0.


def test_many_sources_strides_and_sizes():
    """K = 5..8 (row kernel), unaligned rows (scalar path), views, vs the oracle; size-independent
    properties at a full-size batch."""
    from padertorch_amd.ops import pit_loss
    from padertorch_amd.ops.losses import regression as R
    rng = np.random.RandomState(3)
    for K, T in ((5, 1001), (8, 300), (1, 77), (3, 4099)):
        tgt = rng.randn(K, T).astype(np.float32)
        est = (tgt[::-1] * 0.7 + 0.2 * rng.randn(K, T)).astype(np.float32)
        for name, fn, ofn in (('si-sdr', R.si_sdr_loss, L.td_si_sdr_loss), ('log-mse', R.log_mse_loss, L.td_log_mse_loss)):
            got, perm = pit_loss(dev(est), dev(tgt), axis=0, loss_fn=fn, return_permutation=True)
            want, wperm = L.pit_loss(est, tgt, axis=0, loss_fn=ofn, return_permutation=True)
            assert list(perm) == list(wperm), (K, T, name)
            np.testing.assert_allclose(got.item(), want, rtol=1e-5, atol=1e-5)
    # non-contiguous leading dims / offset views
    big = dev(rng.randn(4, 3, 1030).astype(np.float32))
    e, t = big[1:3, :, 3:1027], big[0:2, :, 5:1029]
    np.testing.assert_allclose(R.sdr_loss(e, t).item(), L.td_sdr_loss(e.cpu().numpy(), t.cpu().numpy()), rtol=1e-5)
    # full-size batch (64 x 2 x 4 s @ 16 kHz): scale invariance, permutation recovery, zero gradient sum
    g = torch.Generator(device='cpu').manual_seed(0)
    s = (0.1 * torch.randn(64, 2, 64000, generator=g)).to(DEV)
    x = (s.flip(1) + 0.01 * torch.randn(64, 2, 64000, generator=g).to(DEV)).requires_grad_(True)
    out = R.pit_td_losses(x, s)
    loss, perm = out['si-sdr']
    assert perm.tolist() == [[1, 0]] * 64
    assert float(loss.detach().max()) < -19. and float(loss.detach().min()) > -21.     # 20 dB by construction
    loss2, _ = R.pit_td_losses(3.7 * x.detach(), s)['si-sdr']
    np.testing.assert_allclose(loss2.cpu().numpy(), loss.detach().cpu().numpy(), atol=1e-4)
    loss.sum().backward()
    # SI-SDR is invariant to scaling the estimate => <grad, estimate> = 0 per row
    inner = (x.grad * x.detach()).sum(-1)
    assert float(inner.abs().max()) < 1e-3 * float((x.grad.abs() * x.detach().abs()).sum(-1).max())


def test_tas_coders(g7):
    from padertorch_amd.contrib.examples.source_separation.tasnet import StftEncoder, IstftDecoder
    mixture = dev(g7['coder/mixture'], True)
    encoded, num_frames = StftEncoder(feature_size=258)(mixture, [203, 150])
    assert list(encoded.shape) == [2, 3, 258, 20] and num_frames.tolist() == g7['coder/num_frames'].tolist()
    np.testing.assert_allclose(encoded.detach().cpu().numpy(), g7['coder/encoded'], atol=1e-5)
    decoded = IstftDecoder(feature_size=258)(dev(g7['coder/stft_signal']))
    assert list(decoded.shape) == [2, 4, 110]
    np.testing.assert_allclose(decoded.cpu().numpy(), g7['coder/decoded'], atol=1e-5)
    enc = StftEncoder(window_length=16, feature_size=66, stride=4)
    dec = IstftDecoder(window_length=16, feature_size=66, stride=4)
    e2 = enc(mixture)
    np.testing.assert_allclose(e2.detach().cpu().numpy(), g7['coder/encoded_16_66_4'], atol=1e-5)
    rt = dec(e2)
    np.testing.assert_allclose(rt.detach().cpu().numpy(), g7['coder/roundtrip_16_66_4'], atol=1e-5)
    # in-graph: the gradient of a quadratic through encoder + decoder matches torch's CPU autograd
    # of the same linear maps (built from the HIP outputs on basis perturbations is too slow; use
    # the adjoint identity <J v, w> = <v, J^T w> instead)
    v = torch.randn_like(mixture)
    w = torch.randn_like(rt)
    (rt * w).sum().backward()
    jtw = mixture.grad
    jv = dec(enc(v))
    np.testing.assert_allclose(float((jv * w).sum()), float((v * jtw).sum()), rtol=2e-4)


def test_edge_shapes():
    """1-D signals, a single source, an empty leading dimension of the mel transform, rank-2 normalisation."""
    from padertorch_amd.ops.losses import regression as R
    from padertorch_amd.contrib.je.modules.features import MelTransform
    from padertorch_amd.modules import normalize
    from oracle import norm_np
    rng = np.random.RandomState(4)
    e, t = rng.randn(333).astype(np.float32), rng.randn(333).astype(np.float32)
    np.testing.assert_allclose(R.si_sdr_loss(dev(e), dev(t)).item(), L.td_si_sdr_loss(e, t), rtol=1e-5)
    np.testing.assert_allclose(R.mse_loss(dev(e), dev(t)).item(), L.td_mse_loss(e, t), rtol=1e-5)
    out = R.pit_td_losses(dev(e[None, None]), dev(t[None, None]), lengths=[300])
    np.testing.assert_allclose(out['log-mse'][0].item(), L.td_log_mse_loss(e[:300], t[:300]), rtol=1e-5)
    assert out['si-sdr'][1].tolist() == [[0]]
    mt = MelTransform(16000, 512, 40).to(DEV)
    assert list(mt(torch.zeros(0, 257, device=DEV)).shape) == [0, 40]
    x = rng.randn(5, 37).astype(np.float32)
    y, m, p, n = normalize(dev(x), None, None, [1], 0, 1, None, True, True, 1e-5)
    want = norm_np.normalize(x, None, None, [1], 0, 1, None, True, True, 1e-5)
    np.testing.assert_allclose(y.cpu().numpy(), want[0], rtol=1e-4, atol=1e-5)
    assert n.cpu().numpy().tolist() == [[37.]] * 5


@pytest.mark.gpu
def test_complex_signals_error_energy_losses():
    """mse / log_mse / log1p_mse / sdr on complex64 signals: the reference's formulas (``regression.py:4-18,47-160``: ``torch.abs``
    of the error, time MEAN over the complex samples) evaluated directly in fp64 on the CPU, values and gradients; the
    scale-invariant losses refuse complex input."""
    import torch
    from padertorch_amd.ops.losses import regression as R
    torch.manual_seed(5)
    e = torch.randn(3, 2, 777, dtype=torch.complex64)
    t = torch.randn(3, 2, 777, dtype=torch.complex64)

    def ref(name, e, t, soft=None):
        err = (e - t).abs() ** 2
        mse = err.mean(-1)
        if name == 'mse':
            return mse.sum()
        if name == 'log_mse':
            x = mse + (10 ** (-soft / 10) * (t.abs() ** 2).mean(-1) if soft else 0)
            return torch.log10(x).sum()
        if name == 'log1p_mse':
            return torch.log10(1 + mse).sum()
        if name == 'sa_sdr':                                 # regression.py:344-392: the squares summed over ALL signals first
            num = (t.abs() ** 2).sum()
            den = err.sum() + (10 ** (-soft / 10) * num if soft else 0)
            return -10 * torch.log10(num / den)
        num = (t.abs() ** 2).sum(-1)
        den = err.sum(-1) + (10 ** (-soft / 10) * num if soft else 0)
        return (-10 * torch.log10(num / den)).mean()

    cases = [('mse', R.mse_loss, {}), ('log_mse', R.log_mse_loss, {}), ('log_mse', R.log_mse_loss, dict(soft_sdr_max=20)),
             ('log1p_mse', R.log1p_mse_loss, {}), ('sdr', R.sdr_loss, {}), ('sdr', R.sdr_loss, dict(soft_sdr_max=30)),
             ('sa_sdr', R.source_aggregated_sdr_loss, {}), ('sa_sdr', R.source_aggregated_sdr_loss, dict(soft_sdr_max=25))]
    for name, fn, kw in cases:
        ed = e.to(torch.complex128).requires_grad_(True)
        want = ref(name, ed, t.to(torch.complex128), kw.get('soft_sdr_max'))
        want.backward()
        eg = e.cuda().requires_grad_(True)
        got = fn(eg, t.cuda(), **kw)
        got.backward()
        assert abs(float(got) - float(want)) < 2e-5 * max(1., abs(float(want))), (name, kw, float(got), float(want))
        assert torch.allclose(eg.grad.cpu().to(torch.complex128), ed.grad, atol=2e-6, rtol=2e-4), (name, kw)


def test_complex_si_sdr_vs_reference(g7):
    """si_sdr_loss on complex64 signals against the REFERENCE's own values and autograd gradients (golden g7 ``complex/*``, generated
    by importing padertorch's regression.py): its scaling factor is the unconjugated product sum(e t) / sum |t|^2 (regression.py:21-24);
    plain, offset invariant, grad_stop, soft threshold, all at once, reduction None."""
    import json
    import torch
    from padertorch_amd.ops.losses import regression as R
    e, t = torch.from_numpy(g7['complex/estimate']), torch.from_numpy(g7['complex/target'])
    for case in json.loads(str(g7['complex/cases'])):
        name, kw = case['name'], case['kwargs']
        eg = e.cuda().requires_grad_(True)
        got = R.si_sdr_loss(eg, t.cuda(), **kw)
        got.sum().backward()
        want, wgrad = g7[f'complex/si_sdr/{name}/value'], g7[f'complex/si_sdr/{name}/grad']
        assert got.shape == want.shape, (name, got.shape, want.shape)
        np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=2e-5, atol=2e-5, err_msg=name)
        np.testing.assert_allclose(eg.grad.cpu().numpy(), wgrad, rtol=2e-4, atol=2e-6 * float(np.abs(wgrad).max() / 1e-3 + 1), err_msg=name)
    # (round 6: the source-aggregated SDR takes complex signals too - test_complex_signals_error_energy_losses has values and gradients)
    e64, t64 = e.to(torch.complex128), t.to(torch.complex128)
    want = -10 * torch.log10((t64.abs() ** 2).sum() / ((e64 - t64).abs() ** 2).sum())
    assert abs(float(R.source_aggregated_sdr_loss(e.cuda(), t.cuda())) - float(want)) < 2e-5 * max(1., abs(float(want)))
