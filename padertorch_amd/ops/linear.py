"""``torch.nn.Linear`` on the split-fp16 GEMM (``csrc/gemm_planes.hip``), with the weight gradient joining the in-place
side-stream accumulation of ``ops.lstm.DEFER_WGRAD``.

The dense layers behind the BLSTM (``padertorch/contrib/examples/source_separation/pit/model.py:98-104``,
``contrib/tcl/dc.py:38-40,62-66``): ``y = x W^T + b`` forward, ``dx = g W`` and ``dW += g^T x``, ``db += sum g``
backward.  With ``DEFER_WGRAD`` (set by the Trainer when it owns flat gradient buffers) the weight gradient is
accumulated straight into ``weight.grad`` on the weight-gradient stream, next to the last BLSTM layer's backward
recurrence instead of in front of it; otherwise it is returned to autograd.  With ``ops.gemm.ENABLED = False`` (or
non-fp32 / CPU tensors) this is ``module(x)``.
"""
import torch

from . import context as _context
from . import gemm as _gemm
from . import lstm as _lstm

__all__ = ['linear']


#: attribute of the output of a fused Linear + ReLU: (tensor version, device word with the float bits of its maximum) - the operand scale
#: the next ``linear`` takes instead of measuring it (the record travels WITH the tensor, like ``ops.lstm.HANDOFF_ATTR``)
AMAX_ATTR = '_ptmi_amax'

#: captured steps: a dense layer's weight gradient is enqueued behind - and starts with - the backward recurrence of the BLSTM layer below it
#: (A/B switch)
DEFER_TO_RECURRENCE = True
#: the same in the eager step (experiment switch, see _LinearFn.backward)
DEFER_IN_EAGER = False
#: ... for weight gradients up to this many flop (2 M N K): a big one beside the recurrence costs more than it frees in front of it
#: (same box, captured step, deferred against not: c2 - 11.7 and 5 GFLOP - 6.581 / 6.616 ms; c3 - 46 and 20 GFLOP - 21.33 / 21.23;
#: c5 - one of 397 GFLOP - 18.87 / 18.11)
DEFER_MAX_FLOP = 15e9

#: ``linear(..., activation='relu')``: the ReLU in the GEMM's epilogue (False: a torch op behind the layer, as before round 4 - A/B switch)
FUSE_RELU = True


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, module, amax_x, relu=False):
        # relu (planes path only, see linear()): y = max(x W^T + b, 0) out of the GEMM's epilogue, which also leaves max y behind
        # (second output, not differentiable); the backward pass masks the incoming gradient and takes ITS maximum in one pass
        amax_w = _gemm.weight_absmax(module.weight)
        amax_x = amax_x if amax_x is not None else _gemm.absmax(x)
        ctx.module = module
        ctx.amax = (amax_x, amax_w)
        ctx.has_bias = bias is not None
        ctx.relu = bool(relu)
        if _gemm.planes_enabled() and x.stride(1) == 1:
            # both operands as fp16 planes (the weight's cached per optimizer step): csrc/gemm_planes.hip
            y = torch.empty((x.shape[0], weight.shape[0]), dtype=torch.float32, device=x.device)
            amax_y = _gemm.zero_word(x.device) if relu else None
            lh = _lstm.handoff_planes_of(x)
            if lh is not None and x.is_contiguous() and x.shape[1] == lh[1] * lh[2]:
                # x is the BLSTM output whose recurrence has left it as fp16 planes of 2^10 h: operand A as it lies
                (scratch, cols), ndir, H = lh
                a, b, K = (scratch, _gemm.scale_word(x.device)), _gemm.weight_planes_h(module.weight, ndir, H, cols), ndir * cols
            else:
                a, b, K = _gemm.pack_n(x, amax_x), _gemm.weight_planes(module.weight), x.shape[1]
            split = _gemm.auto_split_k(x.shape[0], weight.shape[0], K)
            if relu:
                torch.ops.ptmi.gemm_planes_relu_(y, a[0], a[1], b[0], b[1], bias, x.shape[0], weight.shape[0], K, split, amax_y)
                ctx.save_for_backward(x, weight, y)
                ctx.mark_non_differentiable(amax_y)
                ctx.set_materialize_grads(False)        # (no zero-filled "gradient" of the maximum word: a fill launch per step)
                return y, amax_y
            ctx.save_for_backward(x, weight)
            torch.ops.ptmi.gemm_planes_(y, a[0], a[1], b[0], b[1], bias, x.shape[0], weight.shape[0], K, False, split)
            return y
        assert not relu
        ctx.save_for_backward(x, weight)
        return _gemm.mm(x, weight.t(), bias=bias, amax_x=amax_x, amax_y=amax_w)

    @staticmethod
    def backward(ctx, g, _g_amax=None):
        if g is None:
            return None, None, None, None, None, None
        x, weight = ctx.saved_tensors[:2]
        mod = ctx.module
        amax_x, amax_w = ctx.amax
        if ctx.relu:
            amax_g = _gemm.zero_word(g.device)
            g = torch.ops.ptmi.relu_backward_absmax(g if g.stride(1) == 1 else g.contiguous(), ctx.saved_tensors[2], amax_g)
        else:
            g = g.contiguous()
            amax_g = _gemm.absmax(g)
        planes = _gemm.planes_enabled() and x.stride(1) == 1
        if not ctx.needs_input_grad[0]:
            dx = None
        elif planes:
            dx = _gemm.mm_planes_(torch.empty_like(x, memory_format=torch.contiguous_format), _gemm.pack_n(g, amax_g),
                                  _gemm.weight_planes_t(mod.weight), g.shape[0], weight.shape[1], weight.shape[0])
        else:
            dx = _gemm.mm(g, weight, amax_x=amax_g, amax_y=amax_w)
        oc = _context.effective(mod)
        in_place = (oc.defer_wgrad and mod.weight.grad is not None and mod.weight.requires_grad
                    and (not ctx.has_bias or mod.bias.grad is not None))
        if not in_place:
            dw = _gemm.mm(g.t(), x, amax_x=amax_g, amax_y=amax_x) if ctx.needs_input_grad[1] else None
            db = g.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
            return dx, dw, db, None, None, None
        main = torch.cuda.current_stream(x.device)
        side = _lstm._wgrad_stream(x.device) if oc.wgrad_side_stream else main
        has_bias = ctx.has_bias

        def accumulate(start=None):
            if side is not main:
                if start is not None:
                    side.wait_event(start)
                else:
                    side.wait_stream(torch.cuda.current_stream(x.device))
            else:
                main.wait_stream(_lstm._wgrad_stream(x.device))
            with torch.cuda.stream(side):
                if _gemm.planes_enabled() and x.stride(1) == 1:
                    # (beside the top BLSTM layer's backward recurrence: co_resident_split_k)
                    _gemm.mm_planes_(mod.weight.grad, _gemm.pack_t(g, amax_g), _gemm.pack_t(x, amax_x), g.shape[1], x.shape[1],
                                     x.shape[0], accumulate=True, split_k=_gemm.co_resident_split_k(g.shape[1], x.shape[1], x.shape[0]))
                else:
                    _gemm.mm(g.t(), x, out=mod.weight.grad, accumulate=True, amax_x=amax_g, amax_y=amax_x)
                if has_bias:
                    mod.bias.grad.add_(g.sum(0))
            if side is not main:
                for t in (g, x, amax_g) + ((amax_x,) if torch.is_tensor(amax_x) else ()):
                    t.record_stream(side)
            if oc.grad_ready_hook is not None:
                oc.grad_ready_hook([mod.weight] + ([mod.bias] if has_bias else []))

        from . import capture as _capture
        small = 2. * g.shape[0] * g.shape[1] * x.shape[1] <= DEFER_MAX_FLOP
        if DEFER_TO_RECURRENCE and small and (_capture.ACTIVE or DEFER_IN_EAGER) and side is not main and ctx.needs_input_grad[0]:
            # enqueued behind - and started with - the recurrence launch of the BLSTM layer below (ops.lstm.flush_pending_wgrad;
            # sync_deferred enqueues it when there is none).  Started here, linear2's weight gradient ran beside the input-gradient
            # chain relu' -> pack -> GEMM of linear1 that the top layer's backward recurrence waits for (that chain 151 us instead of
            # ~100 in the replay's timeline); a recurrence gives up ~7 % of the time of what runs beside it.  Round 6, one box,
            # alternating: captured c2 step 6.581 against 6.616 ms.  Captured steps only (DEFER_IN_EAGER): in the eager step it is worth
            # 0.015 ms (6.741 against 6.755); gradients bit-identical in one process (scripts/dbg_linear_defer.py) and under a one-rank
            # RCCL group with layer buckets, but the eager BUCKETED two-rank run over gloo (tests/test_gpu_graphed_dp.py: two processes on
            # one GPU) then ends 3e-4 away from the captured one - with linear1's deferred, not with linear2's alone; with or without
            # the start event - not understood, so not shipped.
            _lstm._PENDING_WGRAD.append(accumulate)
        else:
            accumulate()
        return dx, None, None, None, None, None


def linear(module: torch.nn.Linear, x, x_range=None, activation=None):
    """``module(x)`` for a 2-D ``x``.  ``x_range=ops.gemm.UNIT_RANGE`` when ``x`` is known to lie in a range fp16
    covers without scaling (e.g. LSTM outputs); by default its maximum is measured - or taken from the record a fused
    Linear + ReLU has left on ``x``.  ``activation='relu'``: ``relu(module(x))`` (``torch.nn.Linear`` followed by ``torch.nn.ReLU``,
    ``pit/model.py:98-104``) with the activation in the GEMM's epilogue."""
    assert activation in (None, 'relu'), activation
    if x.dim() == 2 and _gemm.usable(x, module.weight):
        oc = _context.effective(module)
        if (oc.grad_use_hook is not None and torch.is_grad_enabled() and oc.defer_wgrad and module.weight.requires_grad
                and module.weight.grad is not None and (module.bias is None or module.bias.grad is not None)):
            oc.grad_use_hook([module.weight] + ([module.bias] if module.bias is not None else []))
        if x_range is None:
            rec = getattr(x, AMAX_ATTR, None)
            if rec is not None and rec[0] == x._version:
                x_range = rec[1]
        if activation == 'relu' and FUSE_RELU and _gemm.PRODUCTS != 1 and _gemm.planes_enabled() and x.stride(1) == 1:
            y, amax_y = _LinearFn.apply(x, module.weight, module.bias, module, x_range, True)
            setattr(y, AMAX_ATTR, (y._version, amax_y))
            return y
        y = _LinearFn.apply(x, module.weight, module.bias, module, x_range)
        return torch.relu(y) if activation == 'relu' else y
    if x.is_cuda:
        from .. import _lib
        _lib.leaving_native_path(f'a Linear({module.in_features}, {module.out_features}) layer',
                                 'ops.gemm.ENABLED = False (library-GEMM A/B mode)' if not _gemm.ENABLED else
                                 f'input of rank {x.dim()} / dtype {x.dtype} (2-D fp32 only)')
    return torch.relu(module(x)) if activation == 'relu' else module(x)
