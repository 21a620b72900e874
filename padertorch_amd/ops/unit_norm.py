"""``torch.nn.functional.normalize(x, dim=-2)`` of the deep-clustering embedding
(``padertorch/contrib/tcl/dc.py:70``) as one HIP pass forward and one backward (``csrc/norm.hip``)."""
import torch

from .. import _lib

__all__ = ['unit_norm']


class _UnitNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        lib = _lib.load()
        N, E, F = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        inv = torch.empty((N, F), dtype=torch.float32, device=x.device)
        _lib.check(_lib.timed('unit_norm_forward', lib.ptmi_unit_norm_forward, _lib.ptr(x), _lib.ptr(y), _lib.ptr(inv),
                              N, E, F, eps, _lib.stream(x.device)), 'ptmi_unit_norm_forward')
        ctx.save_for_backward(y, inv)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, gy):
        y, inv = ctx.saved_tensors
        lib = _lib.load()
        N, E, F = y.shape
        gy = gy.contiguous()
        dx = torch.empty_like(y)
        _lib.check(_lib.timed('unit_norm_backward', lib.ptmi_unit_norm_backward, _lib.ptr(gy), _lib.ptr(y), _lib.ptr(inv),
                              _lib.ptr(dx), N, E, F, ctx.eps, _lib.stream(y.device)), 'ptmi_unit_norm_backward')
        return dx, None


def unit_norm(x, eps=1e-12):
    """``F.normalize(x, p=2, dim=-2, eps=eps)`` for a float32 CUDA tensor ``[N, E, F]`` with ``E <= 32``."""
    _lib.require_gpu(x)
    if x.dim() != 3 or x.dtype != torch.float32 or x.shape[1] > 32:
        raise NotImplementedError('unit_norm: float32 [N, E, F] with E <= 32')
    return _UnitNormFn.apply(x, float(eps))
