"""Per-model switches and hooks of the in-place weight-gradient path ("Threading": the reference's
``Trainer`` may drive one model replica per device from its own host thread, ``padertorch/train/trainer.py:412-420``).

Until round 3 these were module-level names of ``ops.lstm`` that the Trainer set and reset around ``train()``: two Trainers in one
process - a training model and an EMA / validation copy, two models trained alternately, one model per host thread - shared them.
Now a Trainer owns an :class:`OpContext` and attaches it to every module of its model (``attach``); ``ops.lstm.packed_lstm`` and
``ops.linear.linear`` read the context of the module they are given (``effective``).  Fields left at ``None`` inherit the process
defaults, which remain the module-level names (``ops.lstm.DEFER_WGRAD``, ``GRAD_READY_HOOK``, ``GRAD_USE_HOOK``,
``WGRAD_SIDE_STREAM``) for scripts and tests that drive the ops without a Trainer.
"""
from collections import namedtuple

__all__ = ['OpContext', 'attach', 'effective']

_ATTR = '_ptmi_op_context'

Effective = namedtuple('Effective', 'defer_wgrad grad_ready_hook grad_use_hook wgrad_side_stream')


class OpContext:
    """defer_wgrad: accumulate weight gradients in place into ``.grad`` on the side stream (see ``ops.lstm.DEFER_WGRAD``);
    grad_ready_hook(params): called when the in-place gradients of ``params`` are final; grad_use_hook(params): called in the
    forward pass for every use of a module whose gradients will be accumulated in place; wgrad_side_stream: False keeps the
    accumulation on the main stream.  ``None`` = the process default."""
    __slots__ = ('defer_wgrad', 'grad_ready_hook', 'grad_use_hook', 'wgrad_side_stream')

    def __init__(self, defer_wgrad=None, grad_ready_hook=None, grad_use_hook=None, wgrad_side_stream=None):
        self.defer_wgrad = defer_wgrad
        self.grad_ready_hook = grad_ready_hook
        self.grad_use_hook = grad_use_hook
        self.wgrad_side_stream = wgrad_side_stream

    def __repr__(self):
        return 'OpContext(' + ', '.join(f'{k}={getattr(self, k)!r}' for k in self.__slots__) + ')'


def attach(model, context):
    """Make ``context`` the context of every sub-module of ``model`` (``None`` detaches)."""
    for m in model.modules():
        if context is None:
            m.__dict__.pop(_ATTR, None)
        else:
            m.__dict__[_ATTR] = context         # a plain attribute: not a parameter, buffer or sub-module, not in the state_dict
    return context


def effective(module):
    """The values in force for an op applied to ``module``: its attached context's, the process defaults where that says None."""
    from . import lstm as _lstm
    ctx = module.__dict__.get(_ATTR) if module is not None else None
    if ctx is None:
        return Effective(_lstm.DEFER_WGRAD, _lstm.GRAD_READY_HOOK, _lstm.GRAD_USE_HOOK, _lstm.WGRAD_SIDE_STREAM)
    return Effective(
        _lstm.DEFER_WGRAD if ctx.defer_wgrad is None else ctx.defer_wgrad,
        _lstm.GRAD_READY_HOOK if ctx.grad_ready_hook is None else ctx.grad_ready_hook,
        _lstm.GRAD_USE_HOOK if ctx.grad_use_hook is None else ctx.grad_use_hook,
        _lstm.WGRAD_SIDE_STREAM if ctx.wgrad_side_stream is None else ctx.wgrad_side_stream)
