"""Oracle masked normalisation (oracle/norm_np.py) vs the reference goldens g9 (CPU)."""
import numpy as np

from oracle import norm_np as N


def test_normalize_vs_reference(g9):
    for c in g9['cases']:
        k = c['key']
        y, mean, power, n = N.normalize(g9[f'{k}/x'], g9.get(f'{k}/gamma'), g9.get(f'{k}/beta'), c['axes'], c['b_ax'],
                                        c['t_ax'], c['lens'], c['shift'], c['scale'], 1e-3)
        np.testing.assert_allclose(y, g9[f'{k}/y'], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(mean, g9[f'{k}/mean'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(power, g9[f'{k}/power'], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(n, g9[f'{k}/n'])            # counts: exact


def test_doctest_statistics():
    """normalization.py:424-466: x = 2, lengths [1, 2, 3] -> mean 2, power 4, n 6."""
    y, m, p, n = N.normalize(2 * np.ones((3, 10, 4)), None, None, [0, 2], 0, 2, [1, 2, 3], True, True, 1e-3)
    assert (m == 2).all() and (p == 4).all() and (n == 6).all() and m.shape == (1, 10, 1)
    assert (y == 0).all()


def test_running_statistics_and_eval(g9):
    for mod in g9['modules']:
        k = mod['key']
        gamma, beta = g9[f'{k}/gamma'], g9[f'{k}/beta']
        state = dict(num_tracked_values=np.zeros((1, 4, 1)), running_mean=np.zeros((1, 4, 1)),
                     running_power=np.ones((1, 4, 1)))
        for step in range(3):
            x, lens = g9[f'{k}/s{step}/x'], g9[f'{k}/s{step}/lens'].tolist()
            y, mean, power, n = N.normalize(x, gamma, beta, [0, 2], 0, 2, lens, True, True, 1e-5)
            state = N.update_running_stats(state, mean, power, n, mod['momentum'])
            if mod['cls'] == 'inorm':      # InputNormalization: update first, then normalise with the running stats
                y = N.running_norm(x, state, gamma, beta, 0, 2, lens, True, True, 1e-5)
            np.testing.assert_allclose(y, g9[f'{k}/s{step}/y'], rtol=1e-4, atol=1e-4)
            for b in ('num_tracked_values', 'running_mean', 'running_power'):
                np.testing.assert_allclose(state[b], g9[f'{k}/s{step}/{b}'], rtol=1e-5, atol=1e-6)
        y = N.running_norm(g9[f'{k}/eval/x'], state, gamma, beta, 0, 2, [5, 2], True, True, 1e-5)
        np.testing.assert_allclose(y, g9[f'{k}/eval/y'], rtol=1e-4, atol=1e-4)


def test_unit_norm_oracle_matches_torch_normalize():
    """oracle.norm_np.unit_norm / unit_norm_backward == torch.nn.functional.normalize(dim=-2) and its
    autograd (the op of padertorch/contrib/tcl/dc.py:70), incl. all-zero vectors (eps clamp)."""
    import torch
    from oracle import norm_np
    rng = np.random.default_rng(5)
    for shape in [(7, 20, 257), (3, 4, 9), (1, 1, 1), (5, 32, 12)]:
        x = rng.standard_normal(shape).astype(np.float32)
        x[0, :, 0] = 0.0                                   # clamped norm
        g = rng.standard_normal(shape).astype(np.float32)
        xt = torch.tensor(x, requires_grad=True)
        yt = torch.nn.functional.normalize(xt, dim=-2)
        (yt * torch.tensor(g)).sum().backward()
        np.testing.assert_allclose(norm_np.unit_norm(x), yt.detach().numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(norm_np.unit_norm_backward(g, x), xt.grad.numpy(), rtol=2e-5, atol=1e-6)
