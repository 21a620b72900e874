"""Masked normalisation on the MI355X (reference: ``padertorch/modules/normalization.py``).

``Normalization`` / ``InputNormalization`` keep the reference's constructor, buffers
(``num_tracked_values``, ``running_mean``, ``running_power``), parameters (``gamma``, ``beta``),
``forward(x, sequence_lengths)`` semantics, running-statistics update (``:204-216``) and ``inverse``.
``normalize`` mirrors ``normalization.py:414-494``: statistics over ``statistics_axis`` of the
positions ``t < sequence_lengths[b]`` only, optional shift / scale, affine on the independent axes,
masked output; its hand-written backward (``:374-411``) runs as two masked HIP reductions and one
elementwise pass instead of ~25 elementwise torch kernels.

HIP entry points: ``ptmi_norm_reduce`` (masked fp64 sums per group) and ``ptmi_norm_elementwise``.
float32 CUDA(HIP) tensors of rank <= 5 only; there is no CPU fallback.
"""
import torch
from torch import nn
from torch.autograd import Function

from .. import _lib

__all__ = ['Normalization', 'InputNormalization', 'normalize']


def _geometry(shape, statistics_axis, gamma_shape, batch_axis, sequence_axis):
    rank = len(shape)
    if rank > 5:
        raise NotImplementedError(f'rank <= 5 tensors only, got shape {tuple(shape)}')
    stat = {ax % rank for ax in statistics_axis}
    g = _lib.NormGeom()
    g.rank = rank
    sg, ig = 1, 1
    for d in reversed(range(rank)):
        g.size[d] = shape[d]
        if d in stat:
            g.stat_group_stride[d] = 0
        else:
            g.stat_group_stride[d] = sg
            sg *= shape[d]
        if gamma_shape is not None and gamma_shape[d] != 1:
            assert gamma_shape[d] == shape[d], (gamma_shape, shape)
            g.indep_stride[d] = ig
            ig *= shape[d]
        else:
            g.indep_stride[d] = 0
    g.batch_dim = -1 if batch_axis is None else batch_axis % rank
    g.seq_dim = -1 if sequence_axis is None else sequence_axis % rank
    stat_shape = [1 if d in stat else shape[d] for d in range(rank)]
    return g, stat_shape, sg


def _lengths(sequence_lengths, device):
    if sequence_lengths is None:
        return None
    return torch.as_tensor(sequence_lengths, dtype=torch.int32).to(device)


def _reduce(mode, x, gy, lengths, mean, rstd, gamma, geom, shift, n_groups):
    lib = _lib.load()
    ws = torch.empty(int(lib.ptmi_norm_workspace_elems(geom, int(mode == 2))), dtype=torch.float64, device=x.device)
    out = torch.empty((n_groups, 3), dtype=torch.float64, device=x.device)
    _lib.check(_lib.timed(
        'norm_reduce', lib.ptmi_norm_reduce, mode, x.data_ptr(), _lib.ptr(gy), _lib.ptr(lengths), _lib.ptr(mean),
        _lib.ptr(rstd), _lib.ptr(gamma), geom, int(shift), ws.data_ptr(), out.data_ptr(), _lib.stream(x.device)),
        'ptmi_norm_reduce')
    return out


def _elementwise(backward, x, gy, lengths, mean, rstd, gamma, beta, c0, c1, geom, shift, scale):
    lib = _lib.load()
    out = torch.empty_like(x)
    _lib.check(_lib.timed(
        'norm_elementwise', lib.ptmi_norm_elementwise, int(backward), x.data_ptr(), _lib.ptr(gy), _lib.ptr(lengths),
        _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(c0), _lib.ptr(c1), geom,
        int(shift), int(scale), out.data_ptr(), _lib.stream(x.device)), 'ptmi_norm_elementwise')
    return out


def _f32(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def _check_input(x):
    _lib.require_gpu(x)
    if x.dtype != torch.float32:
        raise NotImplementedError(f'float32 only, got {x.dtype}')
    return x.contiguous()


class _Normalize(Function):
    """``normalization.py:322-411`` on the HIP kernels."""

    @staticmethod
    def forward(ctx, x, gamma, beta, statistics_axis, batch_axis, sequence_axis, sequence_lengths, shift, scale,
                eps):
        x = _check_input(x)
        if gamma is not None:
            assert gamma.dim() == x.dim(), gamma.shape
        if beta is not None:
            assert beta.dim() == x.dim(), beta.shape
        affine_shape = gamma.shape if gamma is not None else (beta.shape if beta is not None else None)
        geom, stat_shape, n_groups = _geometry(x.shape, statistics_axis, affine_shape, batch_axis, sequence_axis)
        lengths = _lengths(sequence_lengths, x.device)
        stats = _reduce(0, x, None, lengths, None, None, None, geom, shift, n_groups)
        n_values = stats[:, 2]
        denom = torch.clamp(n_values, min=1)
        mean64 = stats[:, 0] / denom
        power64 = stats[:, 1] / denom
        power_scale = power64 - mean64 ** 2 if shift else power64
        power_scale = torch.clamp(power_scale, min=0.)
        mean = mean64.to(torch.float32)
        rstd = torch.rsqrt(power_scale + eps).to(torch.float32)
        y = _elementwise(False, x, None, lengths, mean if shift else None, rstd if scale else None, _f32(gamma),
                         _f32(beta), None, None, geom, shift, scale)
        ctx.geom, ctx.shift, ctx.scale, ctx.eps, ctx.n_groups = geom, shift, scale, eps, n_groups
        ctx.affine = (gamma is not None, beta is not None, affine_shape)
        ctx.save_for_backward(x, gamma, lengths, mean, rstd, power_scale, denom)
        n_out = n_values.to(torch.float32).reshape(stat_shape)
        ctx.mark_non_differentiable(n_out)
        return y, mean.reshape(stat_shape), power64.to(torch.float32).reshape(stat_shape), n_out

    @staticmethod
    def backward(ctx, grad_y, grad_mean, grad_power, _):
        # like the reference, gradients w.r.t. the returned statistics are not propagated
        x, gamma, lengths, mean, rstd, power_scale, n = ctx.saved_tensors
        geom, shift, scale, eps = ctx.geom, ctx.shift, ctx.scale, ctx.eps
        has_gamma, has_beta, affine_shape = ctx.affine
        grad_y = grad_y.to(torch.float32).contiguous()
        g32 = _f32(gamma)
        a = _reduce(1, x, grad_y, lengths, mean, None, g32, geom, shift, ctx.n_groups)
        sum_g, sum_gx, sum_x = a[:, 0], a[:, 1], a[:, 2]
        scale_ = torch.sqrt(power_scale + eps)
        grad_mean_ = -sum_g if shift else torch.zeros_like(sum_g)
        c1 = torch.zeros_like(sum_g)
        if scale:
            grad_power_ = sum_gx * (-1 / 2) * (power_scale + eps) ** (-3 / 2)
            if shift:
                grad_mean_ = grad_mean_ / scale_ - 2 * grad_power_ * sum_x / n
            c1 = grad_power_ * 2 / n
        c0 = grad_mean_ / n if shift else torch.zeros_like(sum_g)
        grad_x = None
        if ctx.needs_input_grad[0]:
            grad_x = _elementwise(True, x, grad_y, lengths, mean if shift else None, rstd if scale else None, g32,
                                  None, c0.to(torch.float32), c1.to(torch.float32), geom, shift, scale)
        grad_gamma = grad_beta = None
        if has_gamma or has_beta:
            n_indep = 1
            for s in affine_shape:
                n_indep *= s
            # xhat of the gamma gradient is the normalised input: rstd = 1 when scale is off
            rs = rstd if scale else torch.ones_like(rstd)
            p = _reduce(2, x, grad_y, lengths, mean, rs, None, geom, shift, n_indep)
            if has_gamma and ctx.needs_input_grad[1]:
                grad_gamma = p[:, 0].to(torch.float32).reshape(affine_shape)
            if has_beta and ctx.needs_input_grad[2]:
                grad_beta = p[:, 1].to(torch.float32).reshape(affine_shape)
        return grad_x, grad_gamma, grad_beta, None, None, None, None, None, None, None


class _RunningNorm(Function):
    """``Normalization._running_norm`` (``:233-246``): fixed statistics, masked output."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, rstd, statistics_axis, batch_axis, sequence_axis, sequence_lengths):
        x = _check_input(x)
        affine_shape = gamma.shape if gamma is not None else (beta.shape if beta is not None else None)
        geom, _, n_groups = _geometry(x.shape, statistics_axis, affine_shape, batch_axis, sequence_axis)
        lengths = _lengths(sequence_lengths, x.device)
        mean = None if mean is None else _f32(mean).reshape(-1)
        rstd = None if rstd is None else _f32(rstd).reshape(-1)
        shift, scale = mean is not None, rstd is not None
        y = _elementwise(False, x, None, lengths, mean, rstd, _f32(gamma), _f32(beta), None, None, geom, shift, scale)
        ctx.geom, ctx.shift, ctx.scale, ctx.n_groups = geom, shift, scale, n_groups
        ctx.affine = (gamma is not None, beta is not None, affine_shape)
        ctx.save_for_backward(x, gamma, lengths, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, gamma, lengths, mean, rstd = ctx.saved_tensors
        geom, shift, scale = ctx.geom, ctx.shift, ctx.scale
        has_gamma, has_beta, affine_shape = ctx.affine
        grad_y = grad_y.to(torch.float32).contiguous()
        zeros = torch.zeros(ctx.n_groups, dtype=torch.float32, device=x.device)
        grad_x = None
        if ctx.needs_input_grad[0]:
            grad_x = _elementwise(True, x, grad_y, lengths, mean, rstd, _f32(gamma), None, zeros, zeros, geom, shift,
                                  scale)
        grad_gamma = grad_beta = None
        if has_gamma or has_beta:
            n_indep = 1
            for s in affine_shape:
                n_indep *= s
            rs = rstd if scale else torch.ones(ctx.n_groups, dtype=torch.float32, device=x.device)
            p = _reduce(2, x, grad_y, lengths, mean, rs, None, geom, shift, n_indep)
            if has_gamma and ctx.needs_input_grad[1]:
                grad_gamma = p[:, 0].to(torch.float32).reshape(affine_shape)
            if has_beta and ctx.needs_input_grad[2]:
                grad_beta = p[:, 1].to(torch.float32).reshape(affine_shape)
        return grad_x, grad_gamma, grad_beta, None, None, None, None, None, None


def normalize(x, gamma, beta, statistics_axis, batch_axis, sequence_axis, sequence_lengths, shift, scale, eps):
    """``(y, mean, power, n_values)`` like ``normalization.py:414-494``.

    ``x = 2 * ones(3, 10, 4)``, ``sequence_lengths = [1, 2, 3]``, axes ``[0, 2]``: mean 2, power 4, n 6.
    """
    return _Normalize.apply(x, gamma, beta, tuple(statistics_axis), batch_axis, sequence_axis, sequence_lengths,
                            shift, scale, eps)


def _axes(data_format, letters):
    """Positions of the axis ``letters`` in ``data_format`` (both case-insensitive)."""
    fmt = data_format.lower()
    return tuple(fmt.index(a.lower()) for a in letters)


def _stat_shape(shape, rank, keep=None, drop=None):
    """Broadcastable statistics / parameter shape: the sizes of the ``keep`` axes (all others 1), or ``shape`` with the
    ``drop`` axes set to 1.  Every size that stays must be known."""
    if keep is not None:
        out = [shape[ax] if ax in keep else 1 for ax in range(rank)]
    else:
        out = [1 if ax in drop else shape[ax] for ax in range(rank)]
    assert all(d is not None for d in out), (shape, out)
    return out


class Normalization(nn.Module):
    """Masked normalisation layer with the arguments, parameters (``gamma``, ``beta``), buffers
    (``num_tracked_values``, ``running_mean``, ``running_power``) and semantics of the reference class
    (``padertorch/modules/normalization.py:8-262``): statistics over ``statistics_axis`` of the valid part of every
    sequence, learnable scale / shift along ``independent_axis``; when the batch axis is among the statistics axes,
    running statistics are tracked during training (exponentially with ``momentum``, cumulatively with
    ``momentum=None``) and used in evaluation.  The arithmetic runs in ``csrc/norm.hip`` (``normalize`` above)."""

    def __init__(self, data_format='bcft', shape=None, *, statistics_axis='bft', independent_axis='c',
                 batch_axis='b', sequence_axis='t', shift=True, scale=True, eps: float = 1e-5, momentum=0.95):
        super().__init__()
        rank = len(data_format)
        self.data_format = data_format.lower()
        self.batch_axis, = _axes(data_format, batch_axis) if batch_axis is not None else (None,)
        self.sequence_axis, = _axes(data_format, sequence_axis) if sequence_axis is not None else (None,)
        self.statistics_axis = _axes(data_format, statistics_axis)
        self.shift, self.scale, self.eps, self.momentum = shift, scale, eps, momentum
        self.frozen_stats = False
        # running statistics exist iff the statistics are shared across the batch
        self.track_running_stats = batch_axis in statistics_axis
        state = _stat_shape(shape, rank, drop=self.statistics_axis) if self.track_running_stats else None
        for name, wanted, fill in (('num_tracked_values', True, 0.), ('running_mean', shift, 0.), ('running_power', scale, 1.)):
            if state is not None and wanted:
                self.register_buffer(name, torch.full(state, fill))
            else:
                self.register_parameter(name, None)
        self.gamma = self.beta = None
        if independent_axis is not None:
            per_channel = _stat_shape(shape, rank, keep=_axes(data_format, independent_axis))
            if scale:
                self.gamma = nn.Parameter(torch.ones(per_channel))
            if shift:
                self.beta = nn.Parameter(torch.zeros(per_channel))

    @property
    def running_var(self):
        """Unbiased running variance (at least ``eps``) from the tracked power, mean and count."""
        var = self.running_power
        if self.shift:
            n = self.num_tracked_values.clamp(min=2)
            var = var * (n / (n - 1)) - self.running_mean.square()
        return var.clamp(min=0.) + self.eps

    def reset_running_stats(self):
        if not self.track_running_stats:
            return
        self.num_tracked_values.zero_()
        if self.shift:
            self.running_mean.zero_()
        if self.scale:
            self.running_power.fill_(1)

    def _set_trainable(self, flag):
        for p in self.parameters():
            p.requires_grad = flag

    def freeze(self, freeze_stats=True):
        self._set_trainable(False)
        self.frozen_stats = freeze_stats

    def unfreeze(self):
        self._set_trainable(True)
        self.frozen_stats = False

    def forward(self, x, sequence_lengths=None):
        use_batch_statistics = not self.track_running_stats or (self.training and not self.frozen_stats)
        if not use_batch_statistics:
            return self._running_norm(x, sequence_lengths)
        y, mean, power, n_values = normalize(x, self.gamma, self.beta, self.statistics_axis, self.batch_axis,
                                             self.sequence_axis, sequence_lengths, self.shift, self.scale, self.eps)
        if self.track_running_stats:
            self._update_running_stats(mean, power, n_values)
        return y

    def _update_running_stats(self, mean, power, n_values):
        """``running <- m running + (1 - m) new`` with ``m = momentum``, or the share of the values seen before this
        batch when ``momentum`` is ``None`` (cumulative average)."""
        self.num_tracked_values += n_values.detach()
        keep = self.momentum if self.momentum is not None else 1 - n_values / self.num_tracked_values.detach()
        for wanted, running, new in ((self.shift, self.running_mean, mean), (self.scale, self.running_power, power)):
            if wanted:
                running.mul_(keep).add_((1 - keep) * new.detach())

    def _running_norm(self, x, sequence_lengths):
        mean = self.running_mean.detach() if self.shift else None
        rstd = torch.rsqrt(self.running_var.detach() + self.eps) if self.scale else None   # eps enters twice, as in the reference (:238)
        return _RunningNorm.apply(x, self.gamma, self.beta, mean, rstd, self.statistics_axis, self.batch_axis,
                                  self.sequence_axis, sequence_lengths)

    def inverse(self, x, sequence_lengths=None):
        """Undo the running-statistics normalisation (reference ``:248-262``; plain torch ops, not on the training path)."""
        if not self.track_running_stats:
            raise NotImplementedError
        if self.beta is not None:
            x = x - self.beta
        if self.gamma is not None:
            x = x / self.gamma
        if self.scale:
            x = x * torch.sqrt(self.running_var.detach() + self.eps)
        if self.shift:
            x = x + self.running_mean.detach()
        if sequence_lengths is None:
            return x
        from ..ops.sequence.mask import compute_mask
        return x * compute_mask(x, sequence_lengths, self.batch_axis, self.sequence_axis)


class InputNormalization(Normalization):
    """Always normalises with the running statistics when they are tracked (``:265-290``)."""

    def forward(self, x, sequence_lengths=None):
        if self.track_running_stats:
            if self.training:
                with torch.no_grad():
                    xc = _check_input(x)
                    geom, stat_shape, n_groups = _geometry(xc.shape, self.statistics_axis, None, self.batch_axis,
                                                           self.sequence_axis)
                    stats = _reduce(0, xc, None, _lengths(sequence_lengths, xc.device), None, None, None, geom,
                                    self.shift, n_groups)
                    denom = torch.clamp(stats[:, 2], min=1)
                    mean = (stats[:, 0] / denom).to(torch.float32).reshape(stat_shape)
                    power = (stats[:, 1] / denom).to(torch.float32).reshape(stat_shape)
                    self._update_running_stats(mean, power, stats[:, 2].to(torch.float32).reshape(stat_shape))
            x = self._running_norm(x, sequence_lengths)
        else:
            x = super().forward(x, sequence_lengths)
        return x
