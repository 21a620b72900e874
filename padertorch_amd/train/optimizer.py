"""Optimizer wrappers with the interface of ``padertorch/train/optimizer.py:5-90``.

``clip_grad`` / ``step`` / ``zero_grad`` sit on the step path (``trainer.py:512-532``).  The
gradients of all parameters live in ONE flat fp32 bucket (:class:`FlatGrads`): the global-norm clip
is a single reduction over it, ``zero_grad`` a single memset, and the data-parallel exchange a
single RCCL ``all_reduce(SUM)`` (see ``trainer.py``).

On a GPU :class:`Adam` runs the three of them as TWO passes over HBM (``csrc/optim.hip``,
``torch.ops.ptmi.grad_norm`` / ``adam_flat_``): the reproducible 2-norm of the bucket, then clip scale +
moment update + parameter update + zeroing of the bucket in one kernel (32 B per parameter instead of the
~60 B of clip ``mul_`` + multi-tensor Adam + memset).  The moments are views into two flat buffers, the
``torch.optim.Adam`` object keeps owning them (``state_dict`` / ``load_state_dict`` are torch's, i.e. the
reference's checkpoint layout).
"""
import os

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

    def step_and_zero_grad(self):
        """``step()`` then ``zero_grad()`` (``trainer.py:523-524``); one kernel where a subclass can fuse them."""
        self.step()
        self.zero_grad()

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

    # -- device placement / checkpointing of the wrapped torch optimizer (reference API: to / cpu / cuda / state_dict / load_state_dict,
    #    ``optimizer.py:44-70``); the moments live in ``self.optimizer.state``
    def _state_tensors(self):
        self.check_if_set()
        for per_param in self.optimizer.state.values():
            for name, value in per_param.items():
                if torch.is_tensor(value):
                    yield per_param, name, value

    def to(self, device):
        if device is not None:
            for per_param, name, value in list(self._state_tensors()):
                per_param[name] = value.to(device)

    def cpu(self):
        return self.to('cpu')

    def cuda(self, device=None):
        assert device is None or isinstance(device, int), device
        return self.to(torch.device('cuda', device) if isinstance(device, int) else torch.device('cuda'))

    def state_dict(self):
        self.check_if_set()
        return self.optimizer.state_dict()

    def load_state_dict(self, state_dict):
        self.check_if_set()
        return self.optimizer.load_state_dict(state_dict)


class Adam(Optimizer):
    optimizer_cls = optim.Adam
    #: clip + update + zero_grad on the flat bucket by the kernels of ``csrc/optim.hip`` (GPU buckets only)
    native = True

    def __init__(self, gradient_clipping=1e10, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0, amsgrad=False):
        super().__init__(gradient_clipping, lr=lr, betas=betas, eps=eps,
                         weight_decay=weight_decay, amsgrad=amsgrad)
        self._bound = None           # (flat exp_avg, flat exp_avg_sq, steps, segment table, keys) of the native path
        self._norm = None            # 2-norm of the bucket from clip_grad(), applied by the next step()
        #: device scalar (e.g. the sum of the step's losses) whose NON-finiteness makes the next native step skip the update,
        #: next to a non-finite gradient norm; consumed by that step
        self.skip_if_not_finite = None
        #: hyper-parameters as DEVICE words (fp64 [8]: lr, beta1, beta2, eps, weight_decay, gradient_clipping) for steps that are replayed
        #: from a hipGraph (``train.graphed``): kernel arguments are frozen at capture, and the reference's hooks rewrite
        #: ``param_group['lr']`` between iterations (``padertorch/train/hooks.py:736,1029``).  ``use_device_hyper`` switches the native step to
        #: them, ``refresh_device_hyper`` copies the live values in whenever they differ from what the device holds.
        self._hyper = None           # (pinned host fp64 [8], device fp64 [8], the python tuple the device holds)
        self.hyper_from_device = False

    def set_parameters(self, parameters):
        self.parameters = tuple(parameters)
        self._bound = None
        try:        # one fused multi-tensor kernel per step (the model moves to the GPU later)
            self.optimizer = self.optimizer_cls(self.parameters, fused=True, **self.optimizer_kwargs)
        except (RuntimeError, TypeError, ValueError):
            self.optimizer = self.optimizer_cls(self.parameters, **self.optimizer_kwargs)

    # ---------------------------------------------------------------------------------- native path
    def _native_ok(self):
        fg = self.flat_grads
        if not self.native or fg is None or not fg.flat.is_cuda or fg.flat.dtype != torch.float32 or not fg.intact():
            return False
        groups = self.optimizer.param_groups
        if len(groups) != 1:
            return False
        # limits of ptmi_adam_flat (csrc/optim.hip): a segment table of at most kMaxSegs = 1024 contiguous parameter tensors;
        # anything else stays on torch's fused Adam (checked BEFORE _bind rewires the optimizer state)
        if len(fg.params) > 1024 or not all(p.is_contiguous() for p in fg.params):
            return False
        g = groups[0]
        return not (g.get('amsgrad') or g.get('maximize') or g.get('differentiable') or g.get('capturable'))

    def _bind(self):
        """Make every parameter's Adam state a view into flat buffers laid out like the gradient bucket (existing state -
        a loaded checkpoint, earlier torch steps - is copied in)."""
        fg, opt = self.flat_grads, self.optimizer
        keys = tuple(p.data_ptr() for p in fg.params)
        b = self._bound
        if b is not None and b[4] == keys and b[0].device == fg.flat.device and all(
                (st := opt.state.get(p)) is not None and st['exp_avg'].data_ptr() == b[0].data_ptr() + 4 * off
                and st['exp_avg_sq'].data_ptr() == b[1].data_ptr() + 4 * off and st['step'].data_ptr() == b[2].data_ptr() + 4 * i
                for i, (p, off) in enumerate(zip(fg.params, b[5]))):
            return b
        dev = fg.flat.device
        m, v = torch.zeros_like(fg.flat), torch.zeros_like(fg.flat)
        steps = torch.zeros(len(fg.params), dtype=torch.float32, device=dev)
        offs, table, off = [], [], 0
        for i, p in enumerate(fg.params):
            n = p.numel()
            st = opt.state.get(p)
            if st:
                m[off:off + n].copy_(st['exp_avg'].reshape(-1))
                v[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
                steps[i:i + 1].copy_(torch.as_tensor(st['step'], dtype=torch.float32).reshape(1))
            opt.state[p] = {'step': steps[i], 'exp_avg': m[off:off + n].view_as(p), 'exp_avg_sq': v[off:off + n].view_as(p)}
            table.append((p.data_ptr(), off, n))
            offs.append(off)
            off += n
        segs = torch.tensor(table, dtype=torch.int64).to(dev)
        self._bound = (m, v, steps, segs, keys, offs)
        return self._bound

    # ---------------------------------------------------------------------------------- hyper-parameters on the device
    def live_hyper(self):
        """What the next step has to use, read from the places the reference's hooks write (``param_groups``, ``gradient_clipping``)."""
        g = self.optimizer.param_groups[0]
        return (float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['weight_decay']),
                float(self.gradient_clipping))

    def refresh_device_hyper(self, device=None):
        """Make the device words equal the live hyper-parameters (one 64-byte copy on the current stream when they changed, nothing
        otherwise); returns the device tensor.  Stream-ordered: a replay enqueued behind this call reads the new values."""
        live = self.live_hyper()
        if self._hyper is None or (device is not None and self._hyper[1].device != torch.device(device)):
            device = torch.device(device) if device is not None else self.flat_grads.flat.device
            self._hyper = [torch.zeros(8, dtype=torch.float64, pin_memory=True), torch.zeros(8, dtype=torch.float64, device=device), None]
        host, dev, held = self._hyper
        if held != live:
            if held is not None:
                # (the previous copy may still be in flight from these pinned words: a changed value is rare - wait for it)
                torch.cuda.current_stream(dev.device).synchronize()
            host[:6] = torch.tensor(live, dtype=torch.float64)
            dev.copy_(host, non_blocking=True)
            self._hyper[2] = live
        return dev

    def clip_grad(self):
        """Global-norm clipping (``optimizer.py:31-42``); returns the unclipped norm (0-dim tensor).  On the native path
        the scale ``min(1, clip / (norm + 1e-6))`` is applied to the gradients inside the next ``step()``."""
        if not self._native_ok():
            return super().clip_grad()
        self._norm = torch.ops.ptmi.grad_norm(self.flat_grads.flat)
        return self._norm

    @torch.no_grad()
    def _native_step(self, zero_grad):
        m, v, steps, segs, _, _ = self._bind()
        fg, opt = self.flat_grads, self.optimizer
        g = opt.param_groups[0]
        found = getattr(opt, 'found_inf', None)          # device flag set by the Trainer's deferred checks (fused Adam's protocol)
        if found is not None and not (torch.is_tensor(found) and found.is_cuda):
            found = None
        finite, self.skip_if_not_finite = self.skip_if_not_finite, None
        if finite is not None:
            finite = finite.detach().reshape(1).float()
        # (hyper_from_device: a step that is being captured / replayed - the kernel reads lr, betas, eps, weight decay and the clip value
        #  from the device words; whoever replays refreshes them: GraphedStep.__call__)
        hyper = self._hyper[1] if self.hyper_from_device else None
        applied = torch.ops.ptmi.adam_flat_(
            fg.flat, m, v, segs, list(fg.params), self._norm, float(self.gradient_clipping),
            None if found is None else found.reshape(1).float(), finite, steps, float(g['lr']), float(g['betas'][0]),
            float(g['betas'][1]), float(g['eps']), float(g['weight_decay']), bool(zero_grad), hyper)
        # the kernel wrote the parameters through raw pointers: tell autograd (and everything that caches per parameter
        # version, e.g. the operand scales and stacked weights of ops.gemm / ops.lstm) that they changed
        torch.autograd.graph.increment_version(fg.params)
        from ..ops import gemm as _gemm
        _gemm.note_update(fg.params)                      # (the next step's operand forms may start behind THIS kernel)
        steps.add_(applied)                               # a skipped step does not count (fused Adam's semantics)
        self._norm = None

    def step(self):
        self.check_if_set()
        if not self._native_ok():
            return self.optimizer.step()
        return self._native_step(zero_grad=False)

    def step_and_zero_grad(self):
        self.check_if_set()
        if not self._native_ok():
            return super().step_and_zero_grad()
        return self._native_step(zero_grad=True)

    def zero_grad(self):
        self._norm = None
        return super().zero_grad()


class SGD(Optimizer):
    optimizer_cls = optim.SGD

    def __init__(self, gradient_clipping=1e10, lr=1e-3, momentum=0, dampening=0, weight_decay=0,
                 nesterov=False):
        super().__init__(gradient_clipping, lr=lr, momentum=momentum, dampening=dampening,
                         weight_decay=weight_decay, nesterov=nesterov)
