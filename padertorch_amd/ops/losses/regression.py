"""Time-domain regression losses on the MI355X (reference: ``padertorch/ops/losses/regression.py``).

Same functions, arguments, defaults and return values as the reference (``mse_loss`` ``:47-68``,
``log_mse_loss`` ``:71-128``, ``sdr_loss`` ``:131-175``, ``si_sdr_loss`` ``:178-296``,
``log1p_mse_loss`` ``:299-341``, ``source_aggregated_sdr_loss`` ``:344-392``), but every loss is
evaluated from five sums over time per (estimate row, target row) pair,

    See = sum e^2,  Stt = sum t^2,  Set = sum e t,  Se = sum e,  St = sum t,

which ONE HIP streaming pass (``ptmi_td_pair_stats``, fp64 accumulation of exact products) produces
for all K x K pairs of an example at once.  The closed forms below run on the tiny ``[B, K*K+4K]``
float64 statistics tensor with ordinary torch autograd; the gradient w.r.t. the signals is one more
HIP streaming pass (``ptmi_td_lincomb``).  ``pit_loss(..., loss_fn=<one of these>)`` therefore costs one
pass over the signals for all K! permutations (the reference: K! passes per loss function), and
``pit_td_losses`` evaluates several loss functions from the same statistics (the TasNet loss,
``contrib/examples/source_separation/tasnet/model.py:154-176``).

Real float32 CUDA(HIP) tensors only; there is no CPU fallback.
"""
import functools
import itertools
import math

import torch

from ... import _lib

__all__ = [
    'mse_loss',
    'log_mse_loss',
    'sdr_loss',
    'si_sdr_loss',
    'log1p_mse_loss',
    'source_aggregated_sdr_loss',
    'pit_td_losses',
]


class _TdStatsFn(torch.autograd.Function):
    """[B, K, T] x [B, K, T] (time contiguous) -> [B, K*K + 4K] float64 statistics."""

    @staticmethod
    def forward(ctx, est, tgt, lengths):
        lib = _lib.load()
        B, K, T = est.shape
        dev = est.device
        stats = torch.empty((B, int(lib.ptmi_td_stats_elems(K))), dtype=torch.float64, device=dev)
        ws = torch.empty(int(lib.ptmi_td_workspace_elems(B, K, T)), dtype=torch.float64, device=dev)
        strides = _lib.strides4(est.stride(0), est.stride(1), tgt.stride(0), tgt.stride(1))
        _lib.check(_lib.timed(
            'td_pair_stats', lib.ptmi_td_pair_stats, est.data_ptr(), tgt.data_ptr(), _lib.ptr(lengths), B, K, T,
            strides, ws.data_ptr(), stats.data_ptr(), _lib.stream(dev)), 'ptmi_td_pair_stats')
        ctx.save_for_backward(est, tgt, lengths)
        return stats

    @staticmethod
    def backward(ctx, g):
        est, tgt, lengths = ctx.saved_tensors
        lib = _lib.load()
        B, K, T = est.shape
        g = g.to(torch.float32)
        g_set = g[:, :K * K].reshape(B, K, K)
        g_see, g_stt, g_se, g_st = (g[:, K * K + i * K:K * K + (i + 1) * K] for i in range(4))
        st = _lib.stream(est.device)
        grads = [None, None]
        for idx, (x, y, a, bmat, c) in enumerate((
                (est, tgt, 2 * g_see, g_set, g_se),
                (tgt, est, 2 * g_stt, g_set.transpose(1, 2), g_st))):
            if not ctx.needs_input_grad[idx]:
                continue
            out = torch.empty((B, K, T), dtype=torch.float32, device=est.device)
            strides = _lib.strides6(x.stride(0), x.stride(1), y.stride(0), y.stride(1), out.stride(0), out.stride(1))
            a, bmat, c = a.contiguous(), bmat.contiguous(), c.contiguous()
            _lib.check(_lib.timed(
                'td_lincomb', lib.ptmi_td_lincomb, x.data_ptr(), y.data_ptr(), _lib.ptr(lengths), a.data_ptr(),
                bmat.data_ptr(), c.data_ptr(), B, K, T, strides, out.data_ptr(), st), 'ptmi_td_lincomb')
            grads[idx] = out
        return grads[0], grads[1], None


def _as_rows(x, name):
    if not isinstance(x, torch.Tensor):
        raise TypeError(f'{name} must be a torch.Tensor, got {type(x)}')
    _lib.require_gpu(x)
    if x.is_complex():
        raise NotImplementedError('complex signals are not supported by the HIP regression losses')
    if x.dtype != torch.float32:
        raise NotImplementedError(f'float32 signals only, got {x.dtype}')
    return x if x.stride(-1) == 1 or x.shape[-1] == 1 else x.contiguous()


def pair_stats(estimate, target, lengths=None):
    """Statistics of ``[B, K, T]`` signals: dict of float64 tensors ``set [B,K,K]`` (estimate i x
    target j), ``see``, ``stt``, ``se``, ``st`` ``[B,K]`` and ``n [B,1]`` (valid samples)."""
    estimate, target = _as_rows(estimate, 'estimate'), _as_rows(target, 'target')
    assert estimate.shape == target.shape and estimate.dim() == 3, (estimate.shape, target.shape)
    B, K, T = estimate.shape
    if lengths is not None:
        lengths = torch.as_tensor(lengths, dtype=torch.int32, device=estimate.device)
        n = lengths.to(torch.float64).clamp(max=T)[:, None]
    else:
        n = torch.full((B, 1), float(T), dtype=torch.float64, device=estimate.device)
    s = _TdStatsFn.apply(estimate, target, lengths)
    kk = K * K
    return dict(set=s[:, :kk].reshape(B, K, K), see=s[:, kk:kk + K], stt=s[:, kk + K:kk + 2 * K],
                se=s[:, kk + 2 * K:kk + 3 * K], st=s[:, kk + 3 * K:], n=n)


def _get_threshold(soft_sdr_max):
    """tau of the thresholded SDR (``regression.py:39-44``)."""
    if soft_sdr_max is None:
        return None
    assert 1 < soft_sdr_max < 50, f'Uncommon value for soft_sdr_max: {soft_sdr_max}'
    return 10 ** (-soft_sdr_max / 10)


def _reduce(array, reduction):
    if reduction is None or reduction == 'none':
        return array
    if reduction == 'sum':
        return torch.sum(array)
    elif reduction == 'mean':
        return torch.mean(array)
    else:
        raise ValueError(f'Unknown reduction: {reduction}. Choose from "sum", "mean".')


# ---- closed forms: see, stt, set_, se, st, n broadcastable float64 tensors -> loss per row ---------
_LOG10 = math.log(10.)


def _rows_mse(q):
    return (q['see'] - 2 * q['set'] + q['stt']) / q['n']


def _rows_log_mse(q, soft_sdr_max=None):
    loss = _rows_mse(q)
    if soft_sdr_max:
        loss = loss + _get_threshold(soft_sdr_max) * (q['stt'] / q['n'])
    return torch.log10(loss)


def _rows_log1p_mse(q):
    return torch.log10(1 + _rows_mse(q))


def _rows_sdr(q, soft_sdr_max=None):
    den = q['see'] - 2 * q['set'] + q['stt']
    if soft_sdr_max is not None:
        den = den + _get_threshold(soft_sdr_max) * q['stt']
    return -10 * torch.log10(q['stt'] / den)


def _rows_si_sdr(q, offset_invariant=False, grad_stop=False, soft_sdr_max=None):
    see, stt, set_ = q['see'], q['stt'], q['set']
    if offset_invariant:        # statistics of the mean-removed signals
        see = see - q['se'] * q['se'] / q['n']
        stt = stt - q['st'] * q['st'] / q['n']
        set_ = set_ - q['se'] * q['st'] / q['n']
    alpha = set_ / stt
    if grad_stop:
        alpha = alpha.detach()
    s_norm = alpha * alpha * stt                       # ||alpha t||^2
    den = see - 2 * alpha * set_ + s_norm              # ||e - alpha t||^2
    if soft_sdr_max is not None:
        den = den + _get_threshold(soft_sdr_max) * s_norm
    return -10 * torch.log10(s_norm / den)


def _complex_as_real(x):
    """complex64 ``[..., T]`` -> float32 ``[..., 2 T]`` (re, im interleaved): ``sum |e|^2``, ``sum |t|^2`` and ``Re sum e conj(t)``
    of the complex signals are the plain sums of squares / the dot product of these real ones."""
    if x.dtype != torch.complex64:
        raise NotImplementedError(f'complex64 signals only, got {x.dtype}')
    return torch.view_as_real(x.contiguous()).flatten(-2)


def _plain(rows_fn, estimate, target, reduction, **kw):
    halve_n = False
    if isinstance(estimate, torch.Tensor) and isinstance(target, torch.Tensor) and (estimate.is_complex() or target.is_complex()):
        # the error-energy losses are defined through |.| (reference ``regression.py:4-18``: torch.abs) and take complex
        # signals; si_sdr_loss has its own complex form (_si_sdr_complex: the reference's unconjugated scaling factor)
        if rows_fn not in (_rows_mse, _rows_log_mse, _rows_log1p_mse, _rows_sdr):
            raise NotImplementedError('complex signals: mse / log_mse / log1p_mse / sdr / si_sdr losses only')
        assert estimate.is_complex() and target.is_complex(), (estimate.dtype, target.dtype)
        estimate, target = _complex_as_real(estimate), _complex_as_real(target)
        halve_n = True                                  # the time mean runs over T complex samples, not 2 T real ones
    estimate, target = _as_rows(estimate, 'estimate'), _as_rows(target, 'target')
    assert estimate.shape == target.shape, (estimate.shape, target.shape)
    lead, T = estimate.shape[:-1], estimate.shape[-1]
    q = pair_stats(estimate.reshape(-1, 1, T), target.reshape(-1, 1, T))
    q = {k: (v.reshape(-1, 1) if k != 'n' else (v / 2 if halve_n else v)) for k, v in q.items()}
    loss = rows_fn(q, **kw).reshape(lead)
    return _reduce(loss, reduction).to(estimate.dtype)


def mse_loss(estimate: torch.Tensor, target: torch.Tensor, reduction: str = 'sum'):
    """Mean over time of the squared error, ``reduction`` over the rows (``regression.py:47-68``).

    >>> estimate = [[1., 2, 3], [4, 5, 6]]; target = [[2., 3, 4], [4, 0, 6]]    -> tensor(9.3333)
    """
    return _plain(_rows_mse, estimate, target, reduction)


def log_mse_loss(estimate: torch.Tensor, target: torch.Tensor, reduction: str = 'sum',
                 soft_sdr_max: float = None):
    """``log10`` of the mse per row (``regression.py:71-128``); the doctest pair gives 0.9208."""
    return _plain(_rows_log_mse, estimate, target, reduction, soft_sdr_max=soft_sdr_max)


def sdr_loss(estimate: torch.Tensor, target: torch.Tensor, reduction: str = 'mean',
             soft_sdr_max: float = None):
    """Negative (scale dependent) SDR / SNR (``regression.py:131-175``); doctest pair: -6.5167."""
    return _plain(_rows_sdr, estimate, target, reduction, soft_sdr_max=soft_sdr_max)


def si_sdr_loss(estimate, target, reduction='mean', offset_invariant=False, grad_stop=False,
                soft_sdr_max: float = None):
    """Negative SI-SDR (``regression.py:178-296``); doctest pair: -10.7099."""
    assert estimate.shape == target.shape, (estimate.shape, target.shape)
    assert len(estimate.shape) >= 1, estimate.shape
    assert len(estimate.shape) == 1 or estimate.shape[-2] < 10, (
        f'Number of speakers should be small (<10, not {estimate.shape[-2]})!')
    if isinstance(estimate, torch.Tensor) and isinstance(target, torch.Tensor) and (estimate.is_complex() or target.is_complex()):
        return _si_sdr_complex(estimate, target, reduction, offset_invariant, grad_stop, soft_sdr_max)
    return _plain(_rows_si_sdr, estimate, target, reduction, offset_invariant=offset_invariant,
                  grad_stop=grad_stop, soft_sdr_max=soft_sdr_max)


def _si_sdr_complex(estimate, target, reduction, offset_invariant, grad_stop, soft_sdr_max):
    """SI-SDR of complex64 signals as the reference defines it (``regression.py:21-24,178-296``): the scaling factor is the
    UNCONJUGATED product ``alpha = sum(e t) / sum |t|^2`` - a complex number -, ``s = alpha t``, loss ``-10 log10(sum |s|^2 / sum |e -
    s|^2)``.  Real and imaginary parts are taken as the two rows of a [2, T] real signal: ONE pass of ``ptmi_td_pair_stats`` gives
    the four real products ``sum er tr, sum er ti, sum ei tr, sum ei ti`` (its 2 x 2 pair matrix) and the energies, everything else
    is scalar arithmetic per signal under autograd (the gradient w.r.t. the signals is one more streaming pass)."""
    assert estimate.is_complex() and target.is_complex(), (estimate.dtype, target.dtype)
    lead, T = estimate.shape[:-1], estimate.shape[-1]

    def parts(x):       # complex64 [..., T] -> float32 [rows, 2, T] (real row, imaginary row)
        if x.dtype != torch.complex64:
            raise NotImplementedError(f'complex64 signals only, got {x.dtype}')
        return torch.view_as_real(x.contiguous()).reshape(-1, T, 2).transpose(1, 2).contiguous()
    q = pair_stats(parts(estimate), parts(target))
    n = q['n'][:, 0]
    a, c, d, b = q['set'][:, 0, 0], q['set'][:, 0, 1], q['set'][:, 1, 0], q['set'][:, 1, 1]      # er tr, er ti, ei tr, ei ti
    see, stt = q['see'].sum(1), q['stt'].sum(1)
    if offset_invariant:        # statistics of the signals minus their (complex) time means
        (ser, sei), (str_, sti) = q['se'].unbind(1), q['st'].unbind(1)
        a, b, c, d = a - ser * str_ / n, b - sei * sti / n, c - ser * sti / n, d - sei * str_ / n
        see = see - (ser * ser + sei * sei) / n
        stt = stt - (str_ * str_ + sti * sti) / n
    al_re, al_im = (a - b) / stt, (c + d) / stt              # alpha = sum(e t) / sum |t|^2
    if grad_stop:
        al_re, al_im = al_re.detach(), al_im.detach()
    s_norm = (al_re * al_re + al_im * al_im) * stt           # sum |alpha t|^2
    cross = al_re * (a + b) + al_im * (d - c)                # Re(conj(alpha) sum(e conj(t))) = Re sum(e conj(s))
    den = see - 2 * cross + s_norm                           # sum |e - s|^2
    if soft_sdr_max is not None:
        den = den + _get_threshold(soft_sdr_max) * s_norm
    loss = (-10 * torch.log10(s_norm / den)).reshape(lead)
    return _reduce(loss, reduction).to(torch.float32)


def log1p_mse_loss(estimate: torch.Tensor, target: torch.Tensor, reduction: str = 'sum'):
    """``log10(1 + mse)`` per row (``regression.py:299-341``); doctest pair: 1.2711."""
    return _plain(_rows_log1p_mse, estimate, target, reduction)


def source_aggregated_sdr_loss(estimate: torch.Tensor, target: torch.Tensor,
                               soft_sdr_max: float = None) -> torch.Tensor:
    """SDR of the squares summed over ALL rows (``regression.py:344-392``); doctest pair: -4.6133.  complex64 signals: the reference
    takes ``torch.abs`` first (``_sqnorm``, ``regression.py:4-10``) - the sums of squares of the (re, im) rows."""
    if isinstance(estimate, torch.Tensor) and isinstance(target, torch.Tensor) and (estimate.is_complex() or target.is_complex()):
        assert estimate.is_complex() and target.is_complex(), (estimate.dtype, target.dtype)
        estimate, target = _complex_as_real(estimate), _complex_as_real(target)
    estimate, target = _as_rows(estimate, 'estimate'), _as_rows(target, 'target')
    assert estimate.shape == target.shape, (estimate.shape, target.shape)
    T = estimate.shape[-1]
    q = pair_stats(estimate.reshape(-1, 1, T), target.reshape(-1, 1, T))
    return _aggregated(q['see'].sum(), q['stt'].sum(), q['set'].sum(), soft_sdr_max).to(estimate.dtype)


def _aggregated(see, stt, set_, soft_sdr_max):
    den = see - 2 * set_ + stt
    if soft_sdr_max is not None:
        den = den + _get_threshold(soft_sdr_max) * stt
    return -10 * torch.log10(stt / den)


# ---- permutation-invariant use ---------------------------------------------------------------------
#: loss function -> (rows closed form, default reduction)
_KNOWN = {
    mse_loss: (_rows_mse, 'sum'),
    log_mse_loss: (_rows_log_mse, 'sum'),
    sdr_loss: (_rows_sdr, 'mean'),
    si_sdr_loss: (_rows_si_sdr, 'mean'),
    log1p_mse_loss: (_rows_log1p_mse, 'sum'),
}


def resolve(loss_fn):
    """``(rows_fn, reduction, kwargs)`` when ``loss_fn`` is one of this module's losses (possibly
    wrapped in ``functools.partial`` with keyword arguments), else None."""
    kwargs = {}
    while isinstance(loss_fn, functools.partial):
        if loss_fn.args:
            return None
        kwargs = {**loss_fn.keywords, **kwargs}
        loss_fn = loss_fn.func
    if loss_fn is source_aggregated_sdr_loss:
        return 'aggregated', None, kwargs
    if loss_fn not in _KNOWN:
        return None
    rows_fn, reduction = _KNOWN[loss_fn]
    reduction = kwargs.pop('reduction', reduction)
    if reduction not in ('sum', 'mean'):
        return None
    return rows_fn, reduction, kwargs


def pit_from_stats(q, loss_fn):
    """Permutation-invariant loss per example from pairwise statistics (``pit_loss``,
    ``source_separation.py:110-119``: candidates in ``itertools.permutations`` order, the first
    minimum wins).  Returns ``(loss [B] float64, permutation [B, K] int64)`` with
    ``permutation[b][j]`` = estimate row paired with target row ``j``."""
    spec = resolve(loss_fn)
    assert spec is not None, loss_fn
    rows_fn, reduction, kwargs = spec
    B, K = q['see'].shape
    dev = q['see'].device
    perms = list(itertools.permutations(range(K)))
    pidx = torch.tensor(perms, dtype=torch.int64, device=dev)                 # [P, K]
    tj = torch.arange(K, device=dev)
    # pair (estimate perm[j], target j)
    qp = dict(
        set=q['set'][:, pidx, tj],                                            # [B, P, K]
        see=q['see'][:, pidx],
        stt=q['stt'][:, None, :].expand(B, len(perms), K),
        se=q['se'][:, pidx],
        st=q['st'][:, None, :].expand(B, len(perms), K),
        n=q['n'][:, :, None],
    )
    if rows_fn == 'aggregated':
        cand = _aggregated(qp['see'].sum(-1), qp['stt'].sum(-1), qp['set'].sum(-1), kwargs.get('soft_sdr_max'))
    else:
        rows = rows_fn(qp, **kwargs)                                          # [B, P, K]
        cand = rows.sum(-1) if reduction == 'sum' else rows.mean(-1)
    best = torch.argmin(cand.detach(), dim=1)                                 # first minimum
    return cand.gather(1, best[:, None])[:, 0], pidx[best]


def pit_td_losses(estimate, target, lengths=None, loss_fns=None):
    """Batched permutation-invariant time-domain losses from ONE pass over the signals.

    Args:
        estimate, target: ``[B, K, T]`` float32 (padded to T), lengths: valid samples per example.
        loss_fns: dict name -> loss function of this module (default: the TasNet trio).
    Returns:
        dict name -> ``(loss [B] float32, permutation [B, K] int64)``.
    """
    if loss_fns is None:
        loss_fns = {'si-sdr': si_sdr_loss, 'log-mse': log_mse_loss, 'log1p-mse': log1p_mse_loss}
    q = pair_stats(estimate, target, lengths)
    out = {}
    for name, fn in loss_fns.items():
        loss, perm = pit_from_stats(q, fn)
        out[name] = (loss.to(estimate.dtype), perm)
    return out
