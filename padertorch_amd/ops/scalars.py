"""The scalar end of a training step without launches of its own.

Between the loss kernel and its backward kernel the reference's bookkeeping (``padertorch/train/trainer.py:567-638``: pick the losses
out of the review, weight and sum them, ``loss.backward()``) costs autograd a handful of one-element kernels on a GPU - ``ones_like`` for
the root gradient, ``zeros`` + ``copy_`` for every ``vector[i]``, a ``cat`` for the values the summary wants - each a ~5 us launch on the
critical path of a 6.6 ms step (``scripts/dbg_ops_between.py``).  Nothing here changes a value:

* :func:`pick` is ``vector[i]`` whose backward returns a cached one-hot vector (times the incoming gradient, or as it is when that
  gradient is :func:`unit_grad`'s cached one) instead of filling a zero vector and copying into it;
* :func:`unit_grad` is the ``1.`` a ``loss.backward()`` starts from, made once per device instead of per step.
"""
import torch

__all__ = ['pick', 'picked_from', 'unit_grad']

_UNITS = {}
_BASIS = {}

#: attribute of a picked scalar: (the vector it was taken from, its index)
PICK_ATTR = '_ptmi_pick'


def unit_grad(loss):
    """The cached 0-dim ``1.`` of ``loss``'s device and dtype (``torch.autograd.backward(loss, unit_grad(loss))`` is
    ``loss.backward()``); None when it would have to be made inside a stream capture (it has to outlive the graph's pool)."""
    key = (loss.device, loss.dtype)
    one = _UNITS.get(key)
    if one is None:
        from . import capture as _capture
        if _capture.ACTIVE or loss.dim() != 0:
            return None
        one = _UNITS[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    return one if loss.dim() == 0 else None


def _is_unit(g):
    one = _UNITS.get((g.device, g.dtype))
    return one is not None and g.dim() == 0 and g.data_ptr() == one.data_ptr()


def _basis(like, n, i):
    key = (like.device, like.dtype, n, i)
    b = _BASIS.get(key)
    if b is None:
        from . import capture as _capture
        b = torch.zeros(n, dtype=like.dtype, device=like.device)
        b[i] = 1
        if not _capture.ACTIVE:         # (a tensor of a graph's pool is no constant outside its replays)
            _BASIS[key] = b
    return b


class _Pick(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vec, i):
        ctx.i, ctx.n = i, vec.shape[0]
        return vec[i]

    @staticmethod
    def backward(ctx, g):
        b = _basis(g, ctx.n, ctx.i)
        # (the cached one-hot vector itself is handed to autograd: it holds further references, so the engine never accumulates
        # into it in place - torch/csrc/autograd/input_buffer.cpp, can_accumulate_inplace)
        return (b if _is_unit(g) else b * g), None


def pick(vec, i):
    """``vec[i]`` of a 1-D tensor (a loss kernel's vector of losses), see the module docstring."""
    assert vec.dim() == 1, vec.shape
    out = _Pick.apply(vec, int(i)) if vec.requires_grad else vec[int(i)]
    setattr(out, PICK_ATTR, (vec, int(i)))
    return out


def picked_from(t):
    """``(vector, index)`` when ``t`` is a :func:`pick`, else None."""
    return getattr(t, PICK_ATTR, None)
