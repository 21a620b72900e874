"""Optimizer wrappers with the interface of ``padertorch/train/optimizer.py:5-90``.

``clip_grad`` / ``step`` / ``zero_grad`` sit on the step path (``trainer.py:512-532``).  The
gradients of all parameters live in ONE flat fp32 bucket (:class:`FlatGrads`): the global-norm clip
is a single reduction over it, ``zero_grad`` a single memset, and the data-parallel exchange a
single RCCL ``all_reduce(SUM)`` (see ``trainer.py``).
"""
import torch
from torch import optim

__all__ = ['Optimizer', 'Adam', 'SGD', 'FlatGrads']


class FlatGrads:
    """One contiguous gradient buffer; every ``p.grad`` is a view into it."""

    def __init__(self, parameters):
        self.params = [p for p in parameters if p.requires_grad]
        assert self.params, 'no trainable parameters'
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params), 'mixed device/dtype'
        self.flat = torch.zeros(sum(p.numel() for p in self.params), device=dev, dtype=dt)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def intact(self):
        return all(p.grad is not None and p.grad.untyped_storage().data_ptr()
                   == self.flat.untyped_storage().data_ptr() for p in self.params)


class Optimizer:
    optimizer_cls = None
    optimizer = None
    parameters = None
    flat_grads = None

    def __init__(self, gradient_clipping, **kwargs):
        self.gradient_clipping = gradient_clipping
        self.optimizer_kwargs = kwargs

    def set_parameters(self, parameters):
        self.parameters = tuple(parameters)
        self.optimizer = self.optimizer_cls(self.parameters, **self.optimizer_kwargs)

    def check_if_set(self):
        assert self.optimizer is not None, \
            'The optimizer is not initialized, call set_parameter before' \
            ' using any of the optimizer functions'

    def use_flat_grads(self):
        """(Re)bind every gradient into one flat bucket on the parameters' current device."""
        self.check_if_set()
        self.flat_grads = FlatGrads(self.parameters)
        return self.flat_grads

    def zero_grad(self):
        self.check_if_set()
        if self.flat_grads is not None and self.flat_grads.intact():
            self.flat_grads.flat.zero_()
            return None
        return self.optimizer.zero_grad(set_to_none=False)

    def step(self):
        self.check_if_set()
        return self.optimizer.step()

    def clip_grad(self):
        """Global-norm clipping (``optimizer.py:31-42``); returns the unclipped norm (0-dim tensor)."""
        self.check_if_set()
        if self.flat_grads is not None and self.flat_grads.intact():
            flat = self.flat_grads.flat
            total_norm = torch.linalg.vector_norm(flat, 2)
            clip_coef = torch.clamp(self.gradient_clipping / (total_norm + 1e-6), max=1.0)
            flat.mul_(clip_coef)
            return total_norm
        return torch.nn.utils.clip_grad_norm_(self.parameters, self.gradient_clipping)

    def to(self, device):
        if device is None:
            return
        self.check_if_set()
        for state in self.optimizer.state.values():
            for k, v in state.items():
                if torch.is_tensor(v):
                    state[k] = v.to(device)

    def cpu(self):
        return self.to('cpu')

    def cuda(self, device=None):
        assert device is None or isinstance(device, int), device
        return self.to(torch.device('cuda') if device is None else device)

    def load_state_dict(self, state_dict):
        self.check_if_set()
        return self.optimizer.load_state_dict(state_dict)

    def state_dict(self):
        self.check_if_set()
        return self.optimizer.state_dict()


class Adam(Optimizer):
    optimizer_cls = optim.Adam

    def __init__(self, gradient_clipping=1e10, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0, amsgrad=False):
        super().__init__(gradient_clipping, lr=lr, betas=betas, eps=eps,
                         weight_decay=weight_decay, amsgrad=amsgrad)

    def set_parameters(self, parameters):
        self.parameters = tuple(parameters)
        try:        # one fused multi-tensor kernel per step (the model moves to the GPU later)
            self.optimizer = self.optimizer_cls(self.parameters, fused=True, **self.optimizer_kwargs)
        except (RuntimeError, TypeError, ValueError):
            self.optimizer = self.optimizer_cls(self.parameters, **self.optimizer_kwargs)


class SGD(Optimizer):
    optimizer_cls = optim.SGD

    def __init__(self, gradient_clipping=1e10, lr=1e-3, momentum=0, dampening=0, weight_decay=0,
                 nesterov=False):
        super().__init__(gradient_clipping, lr=lr, momentum=momentum, dampening=dampening,
                         weight_decay=weight_decay, nesterov=nesterov)
