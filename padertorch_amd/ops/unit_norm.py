"""``torch.nn.functional.normalize(x, dim=-2)`` of the deep-clustering embedding
(``padertorch/contrib/tcl/dc.py:70``) as one HIP pass forward and one backward (``csrc/norm.hip``)."""
import torch

from .. import _lib
from . import library  # noqa: F401  (registers torch.ops.ptmi.*)

__all__ = ['unit_norm']


class _UnitNormFn(torch.autograd.Function):
    """Kernels: ``torch.ops.ptmi.unit_norm_forward`` / ``unit_norm_backward``."""

    @staticmethod
    def forward(ctx, x, eps):
        y, inv = torch.ops.ptmi.unit_norm_forward(x.contiguous(), eps)
        ctx.save_for_backward(y, inv)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, gy):
        y, inv = ctx.saved_tensors
        return torch.ops.ptmi.unit_norm_backward(gy.contiguous(), y, inv, ctx.eps), None


def unit_norm(x, eps=1e-12):
    """``F.normalize(x, p=2, dim=-2, eps=eps)`` for a float32 CUDA tensor ``[N, E, F]`` (any E: register-tiled kernels up to 32,
    a two-read kernel beyond)."""
    _lib.require_gpu(x)
    if x.dim() != 3 or x.dtype != torch.float32:
        raise NotImplementedError('unit_norm: float32 [N, E, F]')
    return _UnitNormFn.apply(x, float(eps))
