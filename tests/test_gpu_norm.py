"""HIP masked normalisation / StatefulLSTM / SimpleMaskEstimator vs the reference goldens g9 and the
oracle (GPU, through the C ABI)."""
import numpy as np
import pytest
import torch

from oracle import norm_np as N

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a, grad=False):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV).requires_grad_(grad)


def test_normalize_outputs_and_grads(g9):
    """The reference's own test (tests/test_modules/test_norm.py:38-71: outputs to 6 decimals, gradients
    to 4) over a grid of formats / axes / shift / scale."""
    from padertorch_amd.modules import normalize
    for c in g9['cases']:
        k = c['key']
        x = dev(g9[f'{k}/x'], True)
        gamma, beta = dev(g9.get(f'{k}/gamma'), True), dev(g9.get(f'{k}/beta'), True)
        y, mean, power, n = normalize(x, gamma, beta, c['axes'], c['b_ax'], c['t_ax'], c['lens'], c['shift'],
                                      c['scale'], 1e-3)
        np.testing.assert_allclose(y.detach().cpu().numpy(), g9[f'{k}/y'], rtol=1e-5, atol=1e-5, err_msg=k)
        np.testing.assert_allclose(mean.detach().cpu().numpy(), g9[f'{k}/mean'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(power.detach().cpu().numpy(), g9[f'{k}/power'], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(n.detach().cpu().numpy(), g9[f'{k}/n'])
        assert mean.shape == g9[f'{k}/mean'].shape
        (y * dev(g9[f'{k}/w'])).sum().backward()
        np.testing.assert_allclose(x.grad.cpu().numpy(), g9[f'{k}/grad_x'], rtol=2e-4, atol=2e-5, err_msg=k)
        if gamma is not None:
            np.testing.assert_allclose(gamma.grad.cpu().numpy(), g9[f'{k}/grad_gamma'], rtol=2e-4, atol=2e-5, err_msg=k)
        if beta is not None:
            np.testing.assert_allclose(beta.grad.cpu().numpy(), g9[f'{k}/grad_beta'], rtol=2e-4, atol=2e-5, err_msg=k)
    with pytest.raises(RuntimeError):          # no CPU fallback
        normalize(torch.ones(2, 3, 4), None, None, [0, 2], 0, 2, None, True, True, 1e-3)


def test_modules_running_statistics_and_eval(g9):
    from padertorch_amd.modules import Normalization, InputNormalization
    for mod in g9['modules']:
        k = mod['key']
        cls = Normalization if mod['cls'] == 'norm' else InputNormalization
        m = cls(data_format='bct', shape=(None, 4, None), statistics_axis='bt', momentum=mod['momentum']).to(DEV)
        with torch.no_grad():
            m.gamma.copy_(dev(g9[f'{k}/gamma']))
            m.beta.copy_(dev(g9[f'{k}/beta']))
        m.train()
        for step in range(3):
            x = dev(g9[f'{k}/s{step}/x'], True)
            y = m(x, g9[f'{k}/s{step}/lens'].tolist())
            np.testing.assert_allclose(y.detach().cpu().numpy(), g9[f'{k}/s{step}/y'], rtol=1e-4, atol=1e-4)
            y.sum().backward()
            np.testing.assert_allclose(x.grad.cpu().numpy(), g9[f'{k}/s{step}/grad_x'], rtol=2e-4, atol=2e-5)
            for b in ('num_tracked_values', 'running_mean', 'running_power'):
                np.testing.assert_allclose(getattr(m, b).cpu().numpy(), g9[f'{k}/s{step}/{b}'], rtol=1e-5, atol=1e-6)
        m.eval()        # (gamma.grad keeps accumulating over the training steps, as in the golden script)
        x = dev(g9[f'{k}/eval/x'], True)
        y = m(x, [5, 2])
        np.testing.assert_allclose(y.detach().cpu().numpy(), g9[f'{k}/eval/y'], rtol=1e-4, atol=1e-4)
        (y ** 2).sum().backward()
        np.testing.assert_allclose(x.grad.cpu().numpy(), g9[f'{k}/eval/grad_x'], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(m.gamma.grad.cpu().numpy(), g9[f'{k}/eval/grad_gamma'], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(m.inverse(y.detach(), [5, 2]).detach().cpu().numpy(), g9[f'{k}/eval/inverse'], rtol=1e-4, atol=1e-4)
    # state_dict layout equals the reference's (buffers + parameters)
    assert set(m.state_dict()) == {'num_tracked_values', 'running_mean', 'running_power', 'gamma', 'beta'}


def test_simple_mask_estimator(g9):
    from padertorch_amd.contrib.examples.speech_enhancement.mask_estimator.model import SimpleMaskEstimator
    me = SimpleMaskEstimator(17, num_units=32, dropout=0.)
    sd = {k[len('me/sd/'):]: torch.from_numpy(v) for k, v in g9.items() if k.startswith('me/sd/')}
    me.load_state_dict(sd, strict=True)          # the reference's checkpoint loads as is
    me.to(DEV).eval()
    batch = {k: dev(g9[f'me/{k}']) for k in ('observation_abs', 'speech_mask_target', 'noise_mask_target')}
    out = me(batch)
    for k in ('speech_mask_prediction', 'noise_mask_prediction'):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), g9[f'me/{k}'], atol=1e-5)
    review = me.review(batch, out)
    np.testing.assert_allclose(review['loss'].item(), g9['me/loss'], rtol=1e-5)
    review['loss'].backward()
    for k, p in me.named_parameters():
        want = g9[f'me/grad/{k}']
        assert np.abs(p.grad.cpu().numpy() - want).max() <= 2e-4 * max(np.abs(want).max(), 1e-3), k
    assert 'images' not in review                      # rendered only on request (a device -> host copy each)
    me.create_snapshot = True
    images = me.review(batch, me(batch))['images']
    assert set(images) >= {'speech_mask', 'observed_stft', 'noise_mask'}
    assert images['speech_mask'].shape == (1, 17, 11) and images['observed_stft'].shape[1:] == (17, 11)
    me.create_snapshot = False


def test_full_size_properties():
    """B x T x F = 64 x 503 x 257 (BASELINE config 3 shape), ragged: masked positions are exactly
    zero, every (b, f) row of valid frames has mean 0 / power 1, the op is idempotent."""
    from padertorch_amd.modules import normalize
    g = torch.Generator().manual_seed(0)
    x = (3 * torch.randn(64, 503, 257, generator=g) + 1.5).to(DEV)
    lens = [503 - 7 * b for b in range(64)]
    y, mean, power, n = normalize(x, None, None, [1], 0, 1, lens, True, True, 1e-8)
    assert n.shape == (64, 1, 257) and n[:, 0, 0].detach().cpu().tolist() == [float(v) for v in lens]
    for b in (5, 17, 63):
        assert float(y[b, lens[b]:].detach().abs().max()) == 0.
        v = y[b, :lens[b]].detach()
        assert float(v.mean(0).abs().max()) < 1e-5 and float(((v ** 2).mean(0) - 1).abs().max()) < 1e-4
    y2 = normalize(y, None, None, [1], 0, 1, lens, True, True, 1e-8)[0]
    np.testing.assert_allclose(y2.detach().cpu().numpy()[::9, ::50], y.detach().cpu().numpy()[::9, ::50], atol=1e-5)
    want = N.normalize(x[5].cpu().numpy()[None], None, None, [1], 0, 1, [lens[5]], True, True, 1e-8)[0]
    np.testing.assert_allclose(y[5].detach().cpu().numpy(), want[0], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(37, 20, 257), (5, 4, 9), (3, 8, 64), (2, 32, 12), (1, 1, 1), (0, 20, 257), (9, 36, 33), (4, 70, 64)])
def test_unit_norm_vs_oracle_and_torch(shape):
    """ops.unit_norm (ptmi_unit_norm_forward / _backward) == F.normalize(dim=-2) of dc.py:70: forward and
    input gradient against the numpy oracle and torch on the CPU, incl. zero vectors (eps clamp), F not a
    multiple of 4, empty input."""
    import torch
    from oracle import norm_np
    from padertorch_amd import ops
    rng = np.random.default_rng(11)
    x = rng.standard_normal(shape).astype(np.float32)
    if x.size:
        x[0, :, 0] = 0.0
    g = rng.standard_normal(shape).astype(np.float32)
    xd = torch.tensor(x, device='cuda:0', requires_grad=True)
    yd = ops.unit_norm(xd)
    (yd * torch.tensor(g, device='cuda:0')).sum().backward()
    xt = torch.tensor(x, requires_grad=True)
    yt = torch.nn.functional.normalize(xt, dim=-2)
    (yt * torch.tensor(g)).sum().backward()
    np.testing.assert_allclose(yd.detach().cpu().numpy(), norm_np.unit_norm(x), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(yd.detach().cpu().numpy(), yt.detach().numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), norm_np.unit_norm_backward(g, x), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xt.grad.numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.gpu
def test_unit_norm_keeps_nan_like_torch():
    """``F.normalize`` clamps the norm with ``clamp_min`` (a NaN norm stays NaN): a NaN entry makes its whole (n, :, f) vector NaN, the
    others are untouched - ``fmaxf(norm, eps)`` would have divided the vector's finite entries by ``eps`` instead."""
    import torch
    from padertorch_amd import ops
    rng = np.random.default_rng(5)
    x = rng.standard_normal((6, 20, 33)).astype(np.float32)
    x[2, 7, 5] = np.nan
    yd = ops.unit_norm(torch.tensor(x, device='cuda:0')).cpu().numpy()
    yt = torch.nn.functional.normalize(torch.tensor(x), dim=-2).numpy()
    assert np.array_equal(np.isnan(yd), np.isnan(yt)) and np.isnan(yt[2, :, 5]).all() and int(np.isnan(yt).sum()) == 20
    np.testing.assert_allclose(yd[~np.isnan(yt)], yt[~np.isnan(yt)], rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
def test_unit_norm_full_size_properties():
    """C5 embedding size (8 x 503 rows x 20 x 257): unit norms, idempotence, scale invariance, and a
    gradient orthogonal to the output (d/dx of a function of x / |x| has no radial component)."""
    import torch
    from padertorch_amd import ops
    torch.manual_seed(0)
    x = torch.randn(8 * 503, 20, 257, device='cuda:0', requires_grad=True)
    y = ops.unit_norm(x)
    n = y.detach().square().sum(1).sqrt()
    assert float((n - 1).abs().max()) < 1e-6
    assert float((ops.unit_norm(y.detach()) - y.detach()).abs().max()) < 1e-6
    assert float((ops.unit_norm(3.7 * x.detach()) - y.detach()).abs().max()) < 1e-6
    g = torch.randn_like(y)
    (y * g).sum().backward()
    radial = (x.grad * x.detach()).sum(1)
    assert float(radial.abs().max()) < 1e-4 * float(x.grad.abs().max()) * float(x.detach().abs().max()) * 20
