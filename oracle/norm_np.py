"""numpy restatement of the masked normalisation.  ORACLE - test infrastructure only.

Follows ``padertorch/modules/normalization.py``: ``mask_and_compute_stats`` (``:497-512``),
``_Normalize.forward`` (``:345-372``) and the running-statistics bookkeeping of ``Normalization``
(``_update_running_stats`` ``:204-216``, ``running_var`` ``:160-169``, ``_running_norm`` ``:233-246``).
The reference's own test (``tests/test_modules/test_norm.py:8-35``, ``normalize_ref``) states the
same forward; gradients are pinned by the goldens (reference autograd), not restated here.
"""
import numpy as np


def compute_mask(shape, sequence_lengths, batch_axis, sequence_axis):
    """``ops/sequence/mask.py:4-73``."""
    if sequence_lengths is None:
        return np.ones(shape)
    nd = len(shape)
    b, t = batch_axis % nd, sequence_axis % nd
    sb = [1] * nd
    sb[b] = -1
    st = [1] * nd
    st[t] = -1
    idx = np.arange(shape[t]).reshape(st)
    return np.broadcast_to((idx < np.asarray(sequence_lengths).reshape(sb)).astype(np.float64), shape)


def normalize(x, gamma, beta, statistics_axis, batch_axis, sequence_axis, sequence_lengths, shift, scale, eps):
    """``(y, mean, power, n_values)`` in float64."""
    x = np.asarray(x, np.float64)
    axes = tuple(a % x.ndim for a in statistics_axis)
    mask = compute_mask(x.shape, sequence_lengths, batch_axis, sequence_axis)
    n_values = mask.sum(axis=axes, keepdims=True)
    xm = x * mask
    denom = np.maximum(n_values, 1)
    mean = xm.sum(axis=axes, keepdims=True) / denom
    power = (xm ** 2).sum(axis=axes, keepdims=True) / denom
    y = xm
    if shift:
        y = y - mean
        power_scale = power - mean ** 2
    else:
        power_scale = power
    power_scale = np.maximum(power_scale, 0.)
    if scale:
        y = y / np.sqrt(power_scale + eps)
    if gamma is not None:
        y = y * np.asarray(gamma, np.float64)
    if beta is not None:
        y = y + np.asarray(beta, np.float64)
    return y * mask, mean, power, n_values


def update_running_stats(state, mean, power, n_values, momentum, shift=True, scale=True):
    """``Normalization._update_running_stats``; ``state`` = dict(num_tracked_values, running_mean, running_power)."""
    state = {k: (None if v is None else np.array(v, np.float64)) for k, v in state.items()}
    state['num_tracked_values'] = state['num_tracked_values'] + n_values
    m = 1 - n_values / state['num_tracked_values'] if momentum is None else momentum
    if shift:
        state['running_mean'] = state['running_mean'] * m + (1 - m) * mean
    if scale:
        state['running_power'] = state['running_power'] * m + (1 - m) * power
    return state


def running_var(state, eps, shift=True):
    n = np.maximum(state['num_tracked_values'], 2)
    v = state['running_power']
    if shift:
        v = n / (n - 1) * v - state['running_mean'] ** 2
    return np.maximum(v, 0.) + eps


def running_norm(x, state, gamma, beta, batch_axis, sequence_axis, sequence_lengths, shift, scale, eps):
    x = np.asarray(x, np.float64)
    if shift:
        x = x - state['running_mean']
    if scale:
        x = x / np.sqrt(running_var(state, eps, shift) + eps)
    if gamma is not None:
        x = x * gamma
    if beta is not None:
        x = x + beta
    return x * compute_mask(x.shape, sequence_lengths, batch_axis, sequence_axis)


def unit_norm(x, eps=1e-12):
    """``torch.nn.functional.normalize(x, p=2, dim=-2, eps)`` as used on the deep-clustering embedding
    (padertorch/contrib/tcl/dc.py:70): x / max(||x||_2 over axis -2, eps)."""
    x = np.asarray(x)
    n = np.sqrt((x.astype(np.float64) ** 2).sum(-2, keepdims=True))
    return (x / np.maximum(n, eps)).astype(x.dtype)


def unit_norm_backward(gy, x, eps=1e-12):
    """d/dx of ``sum(gy * unit_norm(x))`` (autograd of normalize: the clamp passes no gradient)."""
    x64, g64 = np.asarray(x, np.float64), np.asarray(gy, np.float64)
    n = np.sqrt((x64 ** 2).sum(-2, keepdims=True))
    d = np.maximum(n, eps)
    y = x64 / d
    proj = (g64 * y).sum(-2, keepdims=True) * np.where(n > eps, 1.0, 0.0)
    return ((g64 - y * proj) / d).astype(np.asarray(x).dtype)
