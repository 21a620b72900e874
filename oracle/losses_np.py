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
