"""Oracle (numpy restatement of ops/losses/regression.py) vs the reference goldens g7 (CPU)."""
import functools

import numpy as np

from oracle import losses_np as L

VARIANTS = {
    'mse': L.td_mse_loss, 'log-mse': L.td_log_mse_loss, 'log1p-mse': L.td_log1p_mse_loss, 'sdr': L.td_sdr_loss,
    'si-sdr': L.td_si_sdr_loss, 'sa-sdr': L.td_source_aggregated_sdr_loss,
    'log-mse@20': functools.partial(L.td_log_mse_loss, soft_sdr_max=20),
    'sdr@20': functools.partial(L.td_sdr_loss, soft_sdr_max=20),
    'si-sdr@30': functools.partial(L.td_si_sdr_loss, soft_sdr_max=30),
    'si-sdr-oi': functools.partial(L.td_si_sdr_loss, offset_invariant=True),
    'si-sdr-gs': functools.partial(L.td_si_sdr_loss, grad_stop=True),
    'si-sdr-sum': functools.partial(L.td_si_sdr_loss, reduction='sum'),
    'log-mse-mean': functools.partial(L.td_log_mse_loss, reduction='mean'),
}


def test_doctest_answers(g7):
    """The literal answers of the reference doctests (regression.py:60-67,119-124,148-153,207-212,
    333-338,356-361)."""
    de, dt = g7['doc_estimate'], g7['doc_target']
    literal = {'mse': 9.3333, 'log-mse': 0.9208, 'sdr': -6.5167, 'si-sdr': -10.7099, 'log1p-mse': 1.2711,
               'sa-sdr': -4.6133}
    for n, v in literal.items():
        np.testing.assert_allclose(VARIANTS[n](de, dt), v, atol=5e-5)
        np.testing.assert_allclose(g7[f'doc/{n}'], v, atol=5e-5)
    np.testing.assert_allclose(L.td_mse_loss(de, dt, reduction=None), [1.0, 8.3333], atol=5e-5)
    np.testing.assert_allclose(L.td_si_sdr_loss(de, dt, reduction=None), [-18.2391, -3.1806], atol=5e-5)
    np.testing.assert_allclose(L.td_sdr_loss(dt, dt, soft_sdr_max=20), -20., atol=1e-6)
    np.testing.assert_allclose(L.td_si_sdr_loss(dt, dt, soft_sdr_max=20), -20., atol=1e-6)
    for n in g7['names']:
        np.testing.assert_allclose(VARIANTS[n](de, dt), g7[f'doc/{n}'], rtol=2e-6, atol=2e-6)
    for n in ('mse', 'log-mse', 'log1p-mse', 'sdr', 'si-sdr'):
        np.testing.assert_allclose(VARIANTS[n](de, dt, reduction=None), g7[f'doc_none/{n}'], rtol=2e-6, atol=2e-6)


def test_losses_and_pit_vs_reference(g7):
    for key in g7['cases']:
        est, tgt = g7[f'{key}/estimate'], g7[f'{key}/target']
        for n in g7['names']:
            # the float64 run of the reference pins the restatement tightly, its float32 run loosely
            np.testing.assert_allclose(VARIANTS[n](est, tgt), g7[f'{key}/{n}/loss64'], rtol=1e-10, atol=1e-10)
            np.testing.assert_allclose(VARIANTS[n](est, tgt), g7[f'{key}/{n}/loss'], rtol=2e-5, atol=2e-5)
            loss, perm = L.pit_loss(est, tgt, axis=0, loss_fn=VARIANTS[n], return_permutation=True)
            assert list(perm) == list(g7[f'{key}/{n}/pit_perm']), (key, n)
            np.testing.assert_allclose(loss, g7[f'{key}/{n}/pit_loss'], rtol=2e-5, atol=2e-5)


def test_tasnet_loss(g7):
    got = L.tasnet_losses(g7['tas/x'], g7['tas/s'], g7['tas/num_samples'])
    for k in ('si-sdr', 'log-mse', 'log1p-mse'):
        np.testing.assert_allclose(got[k], g7[f'tas/{k}'], rtol=2e-5, atol=2e-5)
