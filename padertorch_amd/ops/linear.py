"""``torch.nn.Linear`` forward whose weight gradient can join the side-stream accumulation of
``ops.lstm.DEFER_WGRAD`` (the dense layers behind the BLSTM, ``pit/model.py:98-104``): their ``dW``
GEMMs then run next to the last BLSTM layer's backward recurrence instead of in front of it.
Same arithmetic as ``F.linear``; the fallback (flag off, no gradient buffers, evaluation) IS ``F.linear``.
"""
import torch

from . import lstm as _lstm

__all__ = ['linear']


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, module):
        ctx.save_for_backward(x, weight)
        ctx.module = module
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        mod = ctx.module
        g = g.contiguous()
        dx = g @ weight if ctx.needs_input_grad[0] else None
        rows, n_out = g.shape
        main = torch.cuda.current_stream(x.device)
        safe = _lstm.WGRAD_SIDE_STREAM and _lstm.gemm_keys_safe((_lstm.wgrad_key(x.shape[1], n_out, rows),))
        side = _lstm._wgrad_stream(x.device) if safe else main
        side.wait_stream(main)
        with torch.cuda.stream(side):
            mod.weight.grad.addmm_(g.t(), x)
            mod.bias.grad.add_(g.sum(0))
        for t in (g, x):
            t.record_stream(side)
        return dx, None, None, None


def linear(module: torch.nn.Linear, x):
    """``module(x)`` for a 2-D ``x``."""
    if (_lstm.DEFER_WGRAD and x.is_cuda and x.dim() == 2 and torch.is_grad_enabled() and module.bias is not None
            and module.weight.grad is not None and module.bias.grad is not None
            and module.weight.requires_grad and x.dtype == torch.float32):
        return _LinearFn.apply(x, module.weight, module.bias, module)
    return module(x)
