"""numpy (float64) restatement of the source-separation losses.  ORACLE - test infrastructure only.

Follows ``padertorch/ops/losses/source_separation.py``:
  * ``deep_clustering_loss``          :13-31
  * ``pit_loss``                      :34-124  (brute force over ``itertools.permutations``)
  * ``compute_pairwise_losses``       :127-241 (mse branch :219-241)
  * ``pit_loss_from_loss_matrix``     :244-312 ('optimal' = scipy Hungarian)
and the PIT model's review ``padertorch/contrib/examples/source_separation/pit/model.py:112-140``.
Pure-python loops: small cases only.
"""
import itertools

import numpy as np


def mse(a, b):
    """``torch.nn.functional.mse_loss`` default reduction: mean over ALL elements."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.mean((a - b) ** 2)


def pit_loss(estimate, target, axis, loss_fn=mse, return_permutation=False):
    """source_separation.py:94-124.  First minimum wins (``torch.min`` on the stacked candidates).

    The returned permutation is the tuple of *estimate* indices in target order.
    """
    estimate = np.asarray(estimate)
    target = np.asarray(target)
    sources = estimate.shape[axis]
    assert sources < 30
    assert estimate.shape == target.shape, (estimate.shape, target.shape)
    perms = list(itertools.permutations(range(sources)))
    cands = [loss_fn(np.take(estimate, p, axis=axis), target) for p in perms]
    idx = int(np.argmin(np.asarray(cands)))  # argmin -> first occurrence, like torch.min
    if return_permutation:
        return cands[idx], perms[idx]
    return cands[idx]


def pairwise_losses(estimate, target, axis, loss_fn=mse):
    """source_separation.py:219-241: ``L[i, j] = loss_fn(estimate_i, target_j)``."""
    estimate = np.asarray(estimate)
    target = np.asarray(target)
    K = estimate.shape[axis]
    out = np.zeros((K, K))
    for i in range(K):
        for j in range(K):
            out[i, j] = loss_fn(np.take(estimate, i, axis=axis), np.take(target, j, axis=axis))
    return out


def pit_loss_from_loss_matrix(matrix, reduction='mean', return_permutation=False):
    """source_separation.py:277-312 with ``algorithm='optimal'``.

    Returns scipy's ``col_ind`` (target index per estimate) - the INVERSE convention of
    :func:`pit_loss`.
    """
    import scipy.optimize
    matrix = np.asarray(matrix)
    row, col = scipy.optimize.linear_sum_assignment(matrix)
    vals = matrix[row, col]
    if reduction == 'mean':
        loss = vals.mean()
    elif reduction == 'sum':
        loss = vals.sum()
    elif reduction is None:
        loss = vals
    else:
        raise ValueError(reduction)
    if return_permutation:
        return loss, col
    return loss


def deep_clustering_loss(x, t):
    """source_separation.py:26-31: (|X'X|_F^2 - 2|X'T|_F^2 + |T'T|_F^2) / N^2."""
    x = np.asarray(x, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)
    N = x.shape[0]
    return (np.sum((x.T @ x) ** 2) - 2 * np.sum((x.T @ t) ** 2) + np.sum((t.T @ t) ** 2)) / N ** 2


def pit_review_losses(masks, Y_abs, X_abs, cos_phase_difference):
    """pit/model.py:117-140: batch means of the MSE and ideal-phase-sensitive PIT losses.

    Args are per-example lists: mask (T,K,F), Y_abs (T,F), X_abs (T,K,F), cos (T,K,F).
    Returns (pit_mse_loss, pit_ips_loss, perms_mse, perms_ips).
    """
    mse_l, ips_l, pm, pi = [], [], [], []
    for mask, y, x, c in zip(masks, Y_abs, X_abs, cos_phase_difference):
        est = np.asarray(mask, dtype=np.float64) * np.asarray(y, dtype=np.float64)[:, None, :]
        l, p = pit_loss(est, np.asarray(x, dtype=np.float64), axis=-2, return_permutation=True)
        mse_l.append(l)
        pm.append(p)
        l, p = pit_loss(est, np.asarray(x, dtype=np.float64) * np.asarray(c, dtype=np.float64),
                        axis=-2, return_permutation=True)
        ips_l.append(l)
        pi.append(p)
    return float(np.mean(mse_l)), float(np.mean(ips_l)), pm, pi


# ---- time-domain regression losses (padertorch/ops/losses/regression.py) ---------------------------
def _reduce(array, reduction):
    """regression.py:27-36."""
    if reduction is None or reduction == 'none':
        return array
    if reduction == 'sum':
        return np.sum(array)
    if reduction == 'mean':
        return np.mean(array)
    raise ValueError(reduction)


def _threshold(soft_sdr_max):
    """regression.py:39-44."""
    return None if soft_sdr_max is None else 10 ** (-soft_sdr_max / 10)


def td_mse_loss(estimate, target, reduction='sum'):
    """regression.py:47-68: mean over time, ``reduction`` over the rows."""
    e, t = np.asarray(estimate, np.float64), np.asarray(target, np.float64)
    return _reduce(np.mean((e - t) ** 2, axis=-1), reduction)


def td_log_mse_loss(estimate, target, reduction='sum', soft_sdr_max=None):
    """regression.py:71-128."""
    e, t = np.asarray(estimate, np.float64), np.asarray(target, np.float64)
    loss = np.mean((e - t) ** 2, axis=-1)
    if soft_sdr_max:
        loss = loss + _threshold(soft_sdr_max) * np.mean(t * t, axis=-1)
    return _reduce(np.log10(loss), reduction)


def td_log1p_mse_loss(estimate, target, reduction='sum'):
    """regression.py:299-341."""
    e, t = np.asarray(estimate, np.float64), np.asarray(target, np.float64)
    return _reduce(np.log10(1 + np.mean((e - t) ** 2, axis=-1)), reduction)


def td_sdr_loss(estimate, target, reduction='mean', soft_sdr_max=None):
    """regression.py:131-175."""
    e, t = np.asarray(estimate, np.float64), np.asarray(target, np.float64)
    target_norm = np.sum(t * t, axis=-1)
    den = np.sum((e - t) ** 2, axis=-1)
    if soft_sdr_max is not None:
        den = den + _threshold(soft_sdr_max) * target_norm
    with np.errstate(divide='ignore', invalid='ignore'):
        return -_reduce(10 * np.log10(target_norm / den), reduction)


def td_si_sdr_loss(estimate, target, reduction='mean', offset_invariant=False, grad_stop=False,
                   soft_sdr_max=None):
    """regression.py:178-296 (``grad_stop`` only changes gradients)."""
    e, t = np.asarray(estimate, np.float64), np.asarray(target, np.float64)
    if offset_invariant:
        e = e - np.mean(e, axis=-1, keepdims=True)
        t = t - np.mean(t, axis=-1, keepdims=True)
    with np.errstate(divide='ignore', invalid='ignore'):
        alpha = np.sum(e * t, axis=-1, keepdims=True) / np.sum(t * t, axis=-1, keepdims=True)
    return td_sdr_loss(e, alpha * t, reduction=reduction, soft_sdr_max=soft_sdr_max)


def td_source_aggregated_sdr_loss(estimate, target, soft_sdr_max=None):
    """regression.py:344-392."""
    e, t = np.asarray(estimate, np.float64), np.asarray(target, np.float64)
    target_norm = np.sum(t * t)
    den = np.sum((e - t) ** 2)
    if soft_sdr_max is not None:
        den = den + _threshold(soft_sdr_max) * target_norm
    return -10 * np.log10(target_norm / den)


TD_LOSSES = {
    'mse': td_mse_loss,
    'log-mse': td_log_mse_loss,
    'log1p-mse': td_log1p_mse_loss,
    'sdr': td_sdr_loss,
    'si-sdr': td_si_sdr_loss,
    'sa-sdr': td_source_aggregated_sdr_loss,
}


def tasnet_losses(x, s, num_samples):
    """TasNet.loss (contrib/examples/source_separation/tasnet/model.py:154-176): per example
    ``pit_loss(estimated[..., :n], target[..., :n], axis=0, loss_fn)`` for si-sdr / log-mse /
    log1p-mse, batch mean."""
    out = {}
    for name in ('si-sdr', 'log-mse', 'log1p-mse'):
        vals = [pit_loss(e[..., :n], t[..., :n], axis=0, loss_fn=TD_LOSSES[name])
                for n, e, t in zip(num_samples, x, s)]
        out[name] = float(np.mean(vals))
    return out
