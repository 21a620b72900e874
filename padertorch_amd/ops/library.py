"""The HIP kernels of the hot path as registered torch custom ops (``torch.ops.ptmi.*``).

Every op is a thin CUDA-dispatch-key implementation over one entry point (or a fixed pair) of the C ABI in
``include/ptmi.h`` / ``libptmi.so``: tensors in, tensors out, launch on torch's current stream.  They carry no
autograd formula of their own: the differentiable operators (``ops.STFT``, ``ops.pit_features``, ``ops.gemm.mm``,
``ops.packed_lstm``, the loss functions) are ``torch.autograd.Function`` s whose forward and backward call these ops, so
a forward kernel and its hand-written adjoint kernel stay paired.  There is no CPU implementation: calling an op
with CPU tensors fails in the dispatcher (``NotImplementedError: ... 'CPU' backend``).

    torch.ops.ptmi.stft_forward            ptmi_stft_forward                  (padertorch/ops/_stft.py:103-174)
    torch.ops.ptmi.istft_forward           ptmi_istft_forward                 (_stft.py:176-263)
    torch.ops.ptmi.pit_features            ptmi_pit_features                  (pit/data.py:49-77)
    torch.ops.ptmi.pit_loss_forward        ptmi_pit_pairwise_sse + _assign    (ops/losses/source_separation.py:34-312)
    torch.ops.ptmi.pit_loss_backward       ptmi_pit_backward
    torch.ops.ptmi.dc_loss_forward / _backward     ptmi_dc_loss_*             (source_separation.py:13-31)
    torch.ops.ptmi.unit_norm_forward / _backward   ptmi_unit_norm_*           (contrib/tcl/dc.py:70)
    torch.ops.ptmi.lstm_recurrence_forward / _backward   ptmi_lstm_*_persistent, falling back to ptmi_lstm_forward / _backward
                                                                              (torch.nn.LSTM in pit/model.py:60-66,97)
    torch.ops.ptmi.absmax                          ptmi_absmax                (operand scale of the dense layers' fp32 operands)
    torch.ops.ptmi.pack_planes_t / _n, torch.ops.ptmi.gemm_planes_   ptmi_pack_planes_t / _n, ptmi_gemm_planes   (nn.LSTM input projections, nn.Linear, all their gradients)
    torch.ops.ptmi.lstm_weight_prep                ptmi_lstm_weight_prep      (the nn.LSTM parameters' operand forms, once per optimizer step)
    torch.ops.ptmi.grad_norm, torch.ops.ptmi.adam_flat_  ptmi_grad_norm, ptmi_adam_flat (train/optimizer.py:27-42, trainer.py:512-532)
"""
import ctypes
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from .. import _lib

_LIBRARY = torch.library.Library('ptmi', 'DEF')


def _register(schema):
    """``schema``: 'name(args) -> ret' in the dispatcher's schema language; the decorated function becomes the CUDA kernel."""
    name = schema.split('(', 1)[0].strip()

    def deco(fn):
        _LIBRARY.define(schema)
        _LIBRARY.impl(name, fn, 'CUDA')
        return getattr(torch.ops.ptmi, name)
    return deco


def _geom(g: List[int]):
    return _lib.StftGeom(*[int(v) for v in g])


# ------------------------------------------------------------------------------------------------ STFT front-end
@_register('stft_forward(Tensor x, Tensor? row_samples, Tensor window, Tensor twiddle, int[] geom, int frames, int layout, '
           'float edge_scale, str timer) -> Tensor')
def stft_forward(x, row_samples, window, twiddle, geom, frames, layout, edge_scale, timer):
    lib = _lib.load()
    rows, T = x.shape
    F = geom[0] // 2 + 1
    shape = (rows, frames, F, 2) if layout == 0 else (rows, frames, 2 * F)
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    g = _geom(geom)
    args = (x.data_ptr(), rows, x.stride(0), T, _lib.ptr(row_samples), window.data_ptr(), twiddle.data_ptr(), g, frames,
            layout, edge_scale, out.data_ptr(), _lib.stream(x.device))
    rc = _lib.timed(timer, lib.ptmi_stft_forward, *args) if timer else lib.ptmi_stft_forward(*args)
    _lib.check(rc, 'ptmi_stft_forward')
    return out


@_register('istft_forward(Tensor spec, Tensor window, Tensor twiddle, int[] geom, int layout, float edge_scale, int cut_left, '
           'int out_samples, str timer) -> Tensor')
def istft_forward(spec, window, twiddle, geom, layout, edge_scale, cut_left, out_samples, timer):
    lib = _lib.load()
    rows, frames = spec.shape[0], spec.shape[1]
    out = torch.empty((rows, max(out_samples, 0)), dtype=torch.float32, device=spec.device)
    if out_samples > 0:
        args = (spec.data_ptr(), rows, frames, None, window.data_ptr(), twiddle.data_ptr(), _geom(geom), layout, edge_scale,
                cut_left, out_samples, out_samples, out.data_ptr(), _lib.stream(spec.device))
        rc = _lib.timed(timer, lib.ptmi_istft_forward, *args) if timer else lib.ptmi_istft_forward(*args)
        _lib.check(rc, 'ptmi_istft_forward')
    return out


@_register('pit_features(Tensor y, Tensor? s, Tensor? num_samples, Tensor window, Tensor twiddle, int[] geom, int frames) '
           '-> (Tensor, Tensor?, Tensor?)')
def pit_features(y, s, num_samples, window, twiddle, geom, frames):
    return _pit_features(y, s, num_samples, window, twiddle, geom, frames, None, None, None)


@_register('pit_features_packed(Tensor y, Tensor? s, Tensor? num_samples, Tensor window, Tensor twiddle, int[] geom, int frames, '
           'Tensor(a!) log1p_packed, Tensor(b!)? log1p_planes, Tensor? packed_offsets) -> (Tensor, Tensor?, Tensor?)')
def pit_features_packed(y, s, num_samples, window, twiddle, geom, frames, log1p_packed, log1p_planes, packed_offsets):
    """``pit_features`` that also writes the first BLSTM layer's input: ``log1p(Y_abs)`` as PackedSequence rows (fp32, and as fp16
    planes in ``log1p_planes`` - zeroed once by the caller, the kernel never writes the padding)."""
    return _pit_features(y, s, num_samples, window, twiddle, geom, frames, log1p_packed, log1p_planes, packed_offsets)


def _pit_features(y, s, num_samples, window, twiddle, geom, frames, log1p_packed, log1p_planes, packed_offsets):
    lib = _lib.load()
    B, N = y.shape
    K = s.shape[1] if s is not None else 0
    F = geom[0] // 2 + 1
    dev = y.device
    Y_abs = torch.empty((B, frames, F), dtype=torch.float32, device=dev)
    X_abs = cos_pd = None
    if K:
        X_abs = torch.empty((B, frames, K, F), dtype=torch.float32, device=dev)
        cos_pd = torch.empty((B, frames, K, F), dtype=torch.float32, device=dev)
    rc = _lib.timed('pit_features', lib.ptmi_pit_features_packed, y.data_ptr(), _lib.ptr(s), B, K, N, N, _lib.ptr(num_samples),
                    window.data_ptr(), twiddle.data_ptr(), _geom(geom), frames, Y_abs.data_ptr(), _lib.ptr(X_abs),
                    _lib.ptr(cos_pd), _lib.ptr(log1p_packed), _lib.ptr(log1p_planes), _lib.ptr(packed_offsets), _lib.stream(dev))
    if rc == -2:
        raise NotImplementedError(f'pit_features: STFT size {geom[0]} / window {geom[2]} beyond what the direct-DFT kernel stages in LDS')
    _lib.check(rc, 'ptmi_pit_features')
    return Y_abs, X_abs, cos_pd


# ------------------------------------------------------------------------------------------------ losses
@_register('pit_loss_forward(Tensor est, Tensor? obs, Tensor tgt, Tensor? scale, Tensor? row_frames, int B, int T, int K, int F, '
           'int[] strides) -> (Tensor, Tensor, Tensor, Tensor)')
def pit_loss_forward(est, obs, tgt, scale, row_frames, B, T, K, F, strides):
    lib = _lib.load()
    dev = est.device
    nvar = 2 if scale is not None else 1
    ws = torch.empty(int(lib.ptmi_pit_workspace_elems(B, T, K, F)), dtype=torch.float64, device=dev)
    sse = torch.empty((B, nvar, K, K), dtype=torch.float64, device=dev)
    st = _lib.stream(dev)
    _lib.check(_lib.timed('pit_pairwise_sse', lib.ptmi_pit_pairwise_sse, est.data_ptr(), _lib.ptr(obs), tgt.data_ptr(),
                          _lib.ptr(scale), B, T, _lib.strides6(*strides), K, F, _lib.ptr(row_frames), ws.data_ptr(),
                          sse.data_ptr(), st), 'ptmi_pit_pairwise_sse')
    loss = torch.empty(nvar, dtype=torch.float32, device=dev)
    perm = torch.empty((B, nvar, K), dtype=torch.int32, device=dev)
    ex_loss = torch.empty((B, nvar), dtype=torch.float32, device=dev)
    _lib.check(lib.ptmi_pit_assign(sse.data_ptr(), B, nvar, K, F, T, _lib.ptr(row_frames), loss.data_ptr(), perm.data_ptr(),
                                   ex_loss.data_ptr(), st), 'ptmi_pit_assign')
    return loss, perm, ex_loss, sse


@_register('pit_loss_backward(Tensor est, Tensor? obs, Tensor tgt, Tensor? scale, Tensor perm, Tensor g_loss, Tensor? row_frames, '
           'int B, int T, int K, int F, int[] strides) -> Tensor')
def pit_loss_backward(est, obs, tgt, scale, perm, g_loss, row_frames, B, T, K, F, strides):
    lib = _lib.load()
    nvar = 2 if scale is not None else 1
    # the kernel writes every (b, t < T) row, zero for the padded frames t >= T_b
    grad = torch.empty_strided(est.shape, est.stride(), dtype=est.dtype, device=est.device)
    _lib.check(_lib.timed('pit_backward', lib.ptmi_pit_backward, est.data_ptr(), _lib.ptr(obs), tgt.data_ptr(), _lib.ptr(scale),
                          perm.data_ptr(), g_loss.data_ptr(), B, T, _lib.strides6(*strides), K, F, nvar, _lib.ptr(row_frames),
                          grad.data_ptr(), _lib.stream(est.device)), 'ptmi_pit_backward')
    return grad


@_register('dc_loss_forward(Tensor x, Tensor t, Tensor? row_frames, int B, int T, int E, int K, int F, int[] strides) '
           '-> (Tensor, Tensor, Tensor)')
def dc_loss_forward(x, t, row_frames, B, T, E, K, F, strides):
    lib = _lib.load()
    dev = x.device
    ws = torch.empty(int(lib.ptmi_dc_workspace_elems(B, T, F)), dtype=torch.float32, device=dev)
    gram = torch.empty((B, 32, 32), dtype=torch.float64, device=dev)
    ex_loss = torch.empty(B, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    _lib.check(_lib.timed('dc_loss_forward', lib.ptmi_dc_loss_forward, x.data_ptr(), t.data_ptr(), B, T, _lib.strides8(*strides),
                          E, K, F, _lib.ptr(row_frames), ws.data_ptr(), gram.data_ptr(), ex_loss.data_ptr(), loss.data_ptr(),
                          _lib.stream(dev)), 'ptmi_dc_loss_forward')
    return loss, ex_loss, gram


@_register('dc_loss_backward(Tensor x, Tensor t, Tensor gram, Tensor g_loss, Tensor? row_frames, int B, int T, int E, int K, int F, '
           'int[] strides, bool zero_fill) -> Tensor')
def dc_loss_backward(x, t, gram, g_loss, row_frames, B, T, E, K, F, strides, zero_fill):
    lib = _lib.load()
    dx = torch.zeros_like(x, memory_format=torch.preserve_format) if zero_fill \
        else torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=x.device)
    _lib.check(_lib.timed('dc_loss_backward', lib.ptmi_dc_loss_backward, x.data_ptr(), t.data_ptr(), gram.data_ptr(),
                          g_loss.data_ptr(), B, T, _lib.strides8(*strides), E, K, F, _lib.ptr(row_frames), dx.data_ptr(),
                          _lib.stream(x.device)), 'ptmi_dc_loss_backward')
    return dx


# ------------------------------------------------------------------------------------------------ dense layers
_ZERO_WORDS = {}


def _zero_word(device):
    """One int32 word that is zero on the current stream: words of a buffer zeroed ONCE (a fill launch per 256 words instead of a
    zeroing launch in front of every reduction: five per training step); a buffer per stream, each word handed out once."""
    from . import capture as _capture
    if _capture.ACTIVE:                 # a captured step: words the graph itself zeroes at every replay
        return _capture.zero_word(device, 256)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    pool = _ZERO_WORDS.get(key)
    if pool is None or pool[1] >= pool[0].numel():
        if len(_ZERO_WORDS) > 16:
            _ZERO_WORDS.clear()
        pool = _ZERO_WORDS[key] = [torch.zeros(256, dtype=torch.int32, device=device), 0]
    pool[1] += 1
    return pool[0][pool[1] - 1:pool[1]]


@_register('absmax(Tensor x, int rows, int cols, int ld) -> Tensor')
def absmax(x, rows, cols, ld):
    out = _zero_word(x.device)
    _lib.check(_lib.load().ptmi_absmax_accumulate(x.data_ptr(), rows, cols, ld, out.data_ptr(), _lib.stream(x.device)), 'ptmi_absmax')
    return out


@_register('lstm_recurrence_backward_range(Tensor gates, Tensor c, Tensor? c0, Tensor dhy, Tensor w_hh_t, Tensor(a!) dg, '
           'Tensor(b!) scratch, Tensor(c!) dc_carry, Tensor bs_dev, Tensor offs_dev, int T, int max_batch, int rows, int H, '
           'int ndir, int s_begin, int s_end, int prefilled=0, Tensor? dc_n=None) -> bool')
def lstm_recurrence_backward_range(gates, c, c0, dhy, w_hh_t, dg, scratch, dc_carry, bs_dev, offs_dev, T, max_batch, rows, H, ndir,
                                   s_begin, s_end, prefilled=0, dc_n=None):
    """The persistent backward recurrence over the processing steps [s_begin, s_end) (``ptmi_lstm_backward_persistent_range``;
    ranges in order, same ``dg`` / ``scratch`` / ``dc_carry``).  False: the launch cannot be resident (nothing was run)."""
    if dc_n is not None:        # + the gradient w.r.t. the final cell state (ptmi_lstm_backward_persistent_states; whole recurrence)
        assert s_begin == 0 and s_end == T and dc_n.shape == (ndir, max_batch, H) and dc_n.is_contiguous(), (s_begin, s_end, dc_n.shape)
        rc = _lib.timed('lstm_backward', _lib.load().ptmi_lstm_backward_persistent_states, gates.data_ptr(), c.data_ptr(), _lib.ptr(c0),
                        dhy.data_ptr(), dc_n.data_ptr(), w_hh_t.data_ptr(), dg.data_ptr(), bs_dev.data_ptr(), offs_dev.data_ptr(),
                        scratch.data_ptr(), dc_carry.data_ptr(), T, max_batch, rows, H, ndir, int(prefilled), _lib.stream(gates.device))
    else:
        rc = _lib.timed('lstm_backward', _lib.load().ptmi_lstm_backward_persistent_range, gates.data_ptr(), c.data_ptr(), _lib.ptr(c0),
                        dhy.data_ptr(), w_hh_t.data_ptr(), dg.data_ptr(), bs_dev.data_ptr(), offs_dev.data_ptr(), scratch.data_ptr(),
                        dc_carry.data_ptr(), T, max_batch, rows, H, ndir, s_begin, s_end, int(prefilled), _lib.stream(gates.device))
    if rc == -2:
        return False
    _lib.check(rc, 'ptmi_lstm_backward_persistent_range')
    return True


@_register('lstm_recurrence_backward_planes(Tensor gates, Tensor c, Tensor? c0, Tensor dhy, Tensor w_hh_t, Tensor(a!)? dg, '
           'Tensor(b!) dg_t, Tensor(c!) scratch, Tensor(d!)? dc_carry, Tensor bs_dev, Tensor offs_dev, int T, int max_batch, int rows, '
           'int H, int ndir, int s_begin, int s_end, int prefilled=0, Tensor? step_masks=None) -> bool')
def lstm_recurrence_backward_planes(gates, c, c0, dhy, w_hh_t, dg, dg_t, scratch, dc_carry, bs_dev, offs_dev, T, max_batch, rows, H, ndir,
                                    s_begin, s_end, prefilled=0, step_masks=None):
    """``ptmi_lstm_backward_persistent_planes``: the persistent backward recurrence over the processing steps [s_begin, s_end) with
    the gate gradients leaving as bf16 planes of ``dgates^T`` (``dg_t``: ``ndir * ptmi_planes_elems(4H, range rows)`` bf16 values, the
    operand of the weight-gradient GEMMs) and, only when ``dg`` is given, as the row-major fp32 tensor too.  False: the launch
    cannot be resident (nothing was run)."""
    need = ndir * int(_lib.load().ptmi_planes_elems(4 * H, (s_end - s_begin) * max_batch))       # this step range's rows
    assert dg_t.dtype == torch.bfloat16 and dg_t.numel() >= need, (dg_t.dtype, dg_t.numel(), need)
    if step_masks is not None:          # a row-slot batch (ptmi_lstm_backward_persistent_slots): the whole recurrence in one launch
        assert c0 is None and s_begin == 0 and s_end == T, 'row-slot batches: no initial states, no step ranges'
        rc = _lib.timed('lstm_backward', _lib.load().ptmi_lstm_backward_persistent_slots, gates.data_ptr(), c.data_ptr(), dhy.data_ptr(),
                        w_hh_t.data_ptr(), _lib.ptr(dg), dg_t.data_ptr(), bs_dev.data_ptr(), offs_dev.data_ptr(), step_masks.data_ptr(),
                        scratch.data_ptr(), T, max_batch, rows, H, ndir, int(prefilled), _lib.stream(gates.device))
    else:
        rc = _lib.timed('lstm_backward', _lib.load().ptmi_lstm_backward_persistent_planes, gates.data_ptr(), c.data_ptr(), _lib.ptr(c0),
                        dhy.data_ptr(), w_hh_t.data_ptr(), _lib.ptr(dg), dg_t.data_ptr(), bs_dev.data_ptr(), offs_dev.data_ptr(),
                        scratch.data_ptr(), _lib.ptr(dc_carry), T, max_batch, rows, H, ndir, s_begin, s_end, int(prefilled),
                        _lib.stream(gates.device))
    if rc == -2:
        return False
    _lib.check(rc, 'ptmi_lstm_backward_persistent_planes')
    return True


@_register('lstm_bias_grad_add_(Tensor db, Tensor(a!)[] bias_ih_grad, Tensor(b!)[] bias_hh_grad) -> ()')
def lstm_bias_grad_add_(db, bias_ih_grad, bias_hh_grad):
    """``bias_ih_grad[d] += db[d]; bias_hh_grad[d] += db[d]`` for every direction in one launch (``ptmi_lstm_bias_grad_add``)."""
    ndir = len(bias_ih_grad)
    n = bias_ih_grad[0].numel()
    assert db.numel() == ndir * n and db.dtype == torch.float32 and db.is_contiguous()
    assert all(t.is_contiguous() and t.numel() == n and t.dtype == torch.float32 for t in list(bias_ih_grad) + list(bias_hh_grad))
    ih = (ctypes.c_void_p * ndir)(*[t.data_ptr() for t in bias_ih_grad])
    hh = (ctypes.c_void_p * ndir)(*[t.data_ptr() for t in bias_hh_grad])
    _lib.check(_lib.load().ptmi_lstm_bias_grad_add(db.data_ptr(), ndir, n, ih, hh, _lib.stream(db.device)), 'ptmi_lstm_bias_grad_add')


# ------------------------------------------------------------------------------------------------ LSTM parameter forms
@_register('lstm_weight_prep(Tensor[] w_ih, Tensor[] w_hh, Tensor[] b_ih, Tensor[] b_hh, int KP) -> '
           '(Tensor, Tensor, Tensor, Tensor, Tensor)')
def lstm_weight_prep(w_ih, w_hh, b_ih, b_hh, KP):
    ndir, (G, I), H = len(w_ih), w_ih[0].shape, w_hh[0].shape[1]
    dev = w_ih[0].device
    Ipad = (I + 3) // 4 * 4
    f32 = dict(dtype=torch.float32, device=dev)
    w_ih_cat, bias = torch.empty((ndir * G, Ipad), **f32), torch.empty(ndir * G, **f32)
    w_pad, w_t = torch.empty((ndir, G, KP), **f32), torch.empty((ndir, H, G), **f32)
    amax = torch.empty(2, dtype=torch.int32, device=dev)
    ptrs = [(ctypes.c_void_p * ndir)(*[t.data_ptr() for t in ts]) for ts in (w_ih, w_hh, b_ih, b_hh)]
    _lib.check(_lib.timed('lstm_weight_prep', _lib.load().ptmi_lstm_weight_prep, *ptrs, ndir, H, I, w_ih_cat.data_ptr(), Ipad,
                          bias.data_ptr(), w_pad.data_ptr(), KP, w_t.data_ptr(), amax.data_ptr(), _lib.stream(dev)),
               'ptmi_lstm_weight_prep')
    return w_ih_cat, bias, w_pad, w_t, amax


# ------------------------------------------------------------------------------------------------ optimizer step
@_register('grad_norm(Tensor flat) -> Tensor')
def grad_norm(flat):
    lib = _lib.load()
    ws = torch.empty(int(lib.ptmi_grad_norm_workspace_elems()), dtype=torch.float64, device=flat.device)
    out = torch.empty((), dtype=torch.float32, device=flat.device)
    _lib.check(_lib.timed('grad_norm', lib.ptmi_grad_norm, flat.data_ptr(), flat.numel(), ws.data_ptr(), out.data_ptr(),
                          _lib.stream(flat.device)), 'ptmi_grad_norm')
    return out


@_register('adam_flat_(Tensor(a!) flat_grad, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, Tensor segments, Tensor(d!)[] params, '
           'Tensor? norm, float max_norm, Tensor? found_inf, Tensor? finite, Tensor step, float lr, float beta1, float beta2, '
           'float eps, float weight_decay, bool zero_grad, Tensor? hyper=None) -> Tensor')
def adam_flat_(flat_grad, exp_avg, exp_avg_sq, segments, params, norm, max_norm, found_inf, finite, step, lr, beta1, beta2, eps,
               weight_decay, zero_grad, hyper=None):
    # `params` are the tensors the segment table points into (listed so that the dispatcher sees what is written);
    # returns the 0-dim fp32 "applied" flag (0: the update was skipped)
    # `hyper`: device fp64 [6] (lr, beta1, beta2, eps, weight_decay, max_norm) that the kernel reads INSTEAD of the float arguments
    if hyper is not None:
        assert hyper.dtype == torch.float64 and hyper.numel() >= 6 and hyper.device == flat_grad.device and hyper.is_contiguous(), hyper
    applied = torch.empty((), dtype=torch.float32, device=flat_grad.device)
    _lib.check(_lib.timed('adam_flat', _lib.load().ptmi_adam_flat, flat_grad.data_ptr(), exp_avg.data_ptr(),
                          exp_avg_sq.data_ptr(), segments.data_ptr(), segments.shape[0], flat_grad.numel(), _lib.ptr(norm),
                          max_norm, _lib.ptr(found_inf), _lib.ptr(finite), applied.data_ptr(), step.data_ptr(), lr, beta1, beta2,
                          eps, weight_decay, _lib.ptr(hyper), int(zero_grad), _lib.stream(flat_grad.device)), 'ptmi_adam_flat')
    return applied


# ------------------------------------------------------------------------------------------------ GEMM on split planes
@_register('pack_planes_t(Tensor x, Tensor? amax) -> Tensor')
def pack_planes_t(x, amax):
    """x [k, c] (fp32, unit inner stride) -> fp16 (hi, lo) planes of the c x k operand (``ptmi_pack_planes_t``)."""
    lib = _lib.load()
    k, c = x.shape
    out = torch.empty(int(lib.ptmi_planes_elems(c, k)), dtype=torch.float16, device=x.device)
    _lib.check(_lib.timed(f'pack_planes_t:{k}x{c}', lib.ptmi_pack_planes_t, x.data_ptr(), k, c, _ld(x), _lib.ptr(amax), out.data_ptr(),
                          _lib.stream(x.device)), 'ptmi_pack_planes_t')
    return out


@_register('pack_planes_n(Tensor x, Tensor? amax) -> Tensor')
def pack_planes_n(x, amax):
    """x [r, k] (fp32, unit inner stride) -> fp16 (hi, lo) planes of the r x k operand (``ptmi_pack_planes_n``)."""
    lib = _lib.load()
    r, k = x.shape
    out = torch.empty(int(lib.ptmi_planes_elems(r, k)), dtype=torch.float16, device=x.device)
    _lib.check(_lib.timed(f'pack_planes_n:{r}x{k}', lib.ptmi_pack_planes_n, x.data_ptr(), r, k, _ld(x), _lib.ptr(amax), out.data_ptr(),
                          _lib.stream(x.device)), 'ptmi_pack_planes_n')
    return out


@_register('gemm_planes_(Tensor(a!) out, Tensor a, Tensor? amax_a, Tensor b, Tensor? amax_b, Tensor? bias, int M, int N, int K, '
           'bool accumulate, int split_k) -> ()')
def gemm_planes_(out, a, amax_a, b, amax_b, bias, M, N, K, accumulate, split_k):
    lib = _lib.load()
    nws = int(lib.ptmi_gemm_planes_workspace_elems(M, N, K, split_k))
    ws = torch.empty(nws, dtype=torch.float32, device=out.device) if nws else None
    _lib.check(_lib.timed(f'gemm_planes:{M}x{N}x{K}:{split_k}', lib.ptmi_gemm_planes, a.data_ptr(), _lib.ptr(amax_a), b.data_ptr(),
                          _lib.ptr(amax_b), _lib.ptr(bias), out.data_ptr(), max(out.stride(0), N), M, N, K, int(accumulate), split_k,
                          _products(), _lib.ptr(ws), _lib.stream(out.device)), 'ptmi_gemm_planes')


@_register('gemm_planes_relu_(Tensor(a!) out, Tensor a, Tensor? amax_a, Tensor b, Tensor? amax_b, Tensor? bias, int M, int N, int K, '
           'int split_k, Tensor(b!) amax_out) -> ()')
def gemm_planes_relu_(out, a, amax_a, b, amax_b, bias, M, N, K, split_k, amax_out):
    """``out = relu(A B^T + bias)`` and the float bits of ``max out`` into the ZEROED word ``amax_out`` (``ptmi_gemm_planes_relu``)."""
    lib = _lib.load()
    nws = int(lib.ptmi_gemm_planes_workspace_elems(M, N, K, split_k))
    ws = torch.empty(nws, dtype=torch.float32, device=out.device) if nws else None
    _lib.check(_lib.timed(f'gemm_planes:{M}x{N}x{K}:{split_k}', lib.ptmi_gemm_planes_relu, a.data_ptr(), _lib.ptr(amax_a), b.data_ptr(),
                          _lib.ptr(amax_b), _lib.ptr(bias), out.data_ptr(), max(out.stride(0), N), M, N, K, split_k, _products(),
                          _lib.ptr(ws), amax_out.data_ptr(), 1, _lib.stream(out.device)), 'ptmi_gemm_planes_relu')


@_register('relu_backward_absmax(Tensor g, Tensor y, Tensor(a!) amax_out) -> Tensor')
def relu_backward_absmax(g, y, amax_out):
    """``g`` where ``y > 0`` (``y``: the ReLU's output) else 0, and the float bits of its ``max |.|`` into the ZEROED word ``amax_out``."""
    lib = _lib.load()
    assert g.dim() == 2 and g.shape == y.shape and g.stride(1) == 1 and y.stride(1) == 1
    out = torch.empty((g.shape[0], g.shape[1]), dtype=torch.float32, device=g.device)
    _lib.check(_lib.timed(f'relu_backward_absmax:{g.shape[0]}x{g.shape[1]}', lib.ptmi_relu_backward_absmax, g.data_ptr(), y.data_ptr(),
                          out.data_ptr(), g.shape[0], g.shape[1], _ld(g), _ld(y), _ld(out), amax_out.data_ptr(), 1, _lib.stream(g.device)),
               'ptmi_relu_backward_absmax')
    return out


def _ld(x):
    """Row stride of a 2-D source with unit inner stride (a single row has none to speak of)."""
    return x.stride(0) if x.shape[0] > 1 else max(x.stride(0), x.shape[1])


def _products():
    """3 (fp32-equivalent) or 1 (the hi planes only: ``ops.gemm.PRODUCTS``, the reduced-precision reporting mode)."""
    from . import gemm
    return 1 if gemm.PRODUCTS == 1 else 3



@_register('pack_planes_bf16(Tensor x, bool transposed) -> Tensor')
def pack_planes_bf16(x, transposed):
    """bf16 (hi, lo) planes, no scale (``ptmi_pack_planes_t_bf16`` / ``_n_bf16``): ``transposed`` - x [k, c], the operand's
    rows are x's columns; else x [r, k]."""
    lib = _lib.load()
    a, b = x.shape
    rows, k = (b, a) if transposed else (a, b)
    out = torch.empty(int(lib.ptmi_planes_elems(rows, k)), dtype=torch.bfloat16, device=x.device)
    fn = lib.ptmi_pack_planes_t_bf16 if transposed else lib.ptmi_pack_planes_n_bf16
    _lib.check(_lib.timed(f'pack_planes_bf16:{a}x{b}', fn, x.data_ptr(), a, b, _ld(x), out.data_ptr(), _lib.stream(x.device)),
               'ptmi_pack_planes_bf16')
    return out


@_register('pack_planes_into_(Tensor(a!) out, Tensor x, Tensor? amax, bool transposed, int kb_total, int kb_offset, int kb_count) -> ()')
def pack_planes_into_(out, x, amax, transposed, kb_total, kb_offset, kb_count):
    """A pack pass writing ``kb_count`` k blocks at k block ``kb_offset`` of planes ``out`` that have ``kb_total`` k blocks per row
    tile (``ptmi_pack_planes_into``); fp16 or bf16 by ``out.dtype``; ``x [r, k]``, or ``[k, c]`` when ``transposed``."""
    lib = _lib.load()
    a, b = x.shape
    rows = b if transposed else a
    assert out.dtype in (torch.float16, torch.bfloat16) and out.numel() >= (rows + 15) // 16 * kb_total * 1024, (out.dtype, out.numel())
    _lib.check(_lib.timed(f'pack_planes_into:{a}x{b}', lib.ptmi_pack_planes_into, x.data_ptr(), a, b, _ld(x), int(transposed),
                          int(out.dtype == torch.bfloat16), _lib.ptr(amax), out.data_ptr(), kb_total, kb_offset, kb_count,
                          _lib.stream(x.device)), 'ptmi_pack_planes_into')


@_register('gemm_planes_bf16_(Tensor(a!) out, Tensor a, int a_offset, Tensor b, Tensor? bias, int M, int N, int K, bool accumulate, '
           'int split_k) -> ()')
def gemm_planes_bf16_(out, a, a_offset, b, bias, M, N, K, accumulate, split_k):
    """``a``: any tensor whose storage holds the bf16 planes of the M x K operand from byte ``a_offset`` on (16-byte aligned) -
    e.g. the scratch of the persistent backward recurrence (``ptmi_lstm_handoff_cols``)."""
    lib = _lib.load()
    nws = int(lib.ptmi_gemm_planes_workspace_elems(M, N, K, split_k))
    ws = torch.empty(nws, dtype=torch.float32, device=out.device) if nws else None
    _lib.check(_lib.timed(f'gemm_planes_bf16:{M}x{N}x{K}:{split_k}', lib.ptmi_gemm_planes_bf16, a.data_ptr() + a_offset, b.data_ptr(),
                          _lib.ptr(bias), out.data_ptr(), max(out.stride(0), N), M, N, K, int(accumulate), split_k, _products(),
                          _lib.ptr(ws), _lib.stream(out.device)), 'ptmi_gemm_planes_bf16')


@_register('gemm_planes_bf16_two_(Tensor(a!) out, Tensor(b!) out2, Tensor a, int a_offset, Tensor b, int M, int N, int K, bool accumulate, '
           'int split_k) -> ()')
def gemm_planes_bf16_two_(out, out2, a, a_offset, b, M, N, K, accumulate, split_k):
    """``[out; out2] (+)= A B^T`` (``ptmi_gemm_planes_bf16_two``): ``out`` takes the first ``out.shape[0]`` rows of the M x N product,
    ``out2`` the rest - two parameters' gradient buffers with one row stride (both directions' ``dW_ih`` of a BLSTM layer)."""
    lib = _lib.load()
    assert out.shape[0] + out2.shape[0] == M and out.shape[1] == out2.shape[1] == N and out.stride(0) == out2.stride(0), (out.shape, out2.shape)
    nws = int(lib.ptmi_gemm_planes_workspace_elems(M, N, K, split_k))
    ws = torch.empty(nws, dtype=torch.float32, device=out.device) if nws else None
    _lib.check(_lib.timed(f'gemm_planes_bf16:{M}x{N}x{K}:{split_k}', lib.ptmi_gemm_planes_bf16_two, a.data_ptr() + a_offset, b.data_ptr(),
                          out.data_ptr(), out2.data_ptr(), out.shape[0], max(out.stride(0), N), M, N, K, int(accumulate), split_k,
                          _products(), _lib.ptr(ws), _lib.stream(out.device)), 'ptmi_gemm_planes_bf16_two')


# ------------------------------------------------------------------------------------------------ unit norm
@_register('unit_norm_forward(Tensor x, float eps) -> (Tensor, Tensor)')
def unit_norm_forward(x, eps):
    N, E, F = x.shape
    y = torch.empty_like(x)
    inv = torch.empty((N, F), dtype=torch.float32, device=x.device)
    _lib.check(_lib.timed('unit_norm_forward', _lib.load().ptmi_unit_norm_forward, _lib.ptr(x), _lib.ptr(y), _lib.ptr(inv),
                          N, E, F, eps, _lib.stream(x.device)), 'ptmi_unit_norm_forward')
    return y, inv


@_register('unit_norm_backward(Tensor gy, Tensor y, Tensor inv, float eps) -> Tensor')
def unit_norm_backward(gy, y, inv, eps):
    N, E, F = y.shape
    dx = torch.empty_like(y)
    _lib.check(_lib.timed('unit_norm_backward', _lib.load().ptmi_unit_norm_backward, _lib.ptr(gy), _lib.ptr(y), _lib.ptr(inv),
                          _lib.ptr(dx), N, E, F, eps, _lib.stream(y.device)), 'ptmi_unit_norm_backward')
    return dx


# ------------------------------------------------------------------------------------------------ (B)LSTM recurrence
@_register('lstm_recurrence_forward(Tensor(a!) gates, Tensor(b!) hy, Tensor? c0, Tensor w_hh_pad, Tensor? w_amax, Tensor bs_dev, '
           'Tensor offs_dev, int bs_host, int offs_host, int T, int max_batch, int rows, int H, int KP, int ndir, bool persistent, '
           'Tensor(c!)? scratch=None, bool prefilled=False, Tensor(d!)? backward_scratch=None, Tensor? step_masks=None) -> (Tensor, Tensor?)')
def lstm_recurrence_forward(gates, hy, c0, w_hh_pad, w_amax, bs_dev, offs_dev, bs_host, offs_host, T, max_batch, rows, H, KP, ndir,
                            persistent, scratch=None, prefilled=False, backward_scratch=None, step_masks=None):
    """gates: pre-activations in, activations out (in place); hy: output rows (a view into the caller's padded buffer).
    Returns (c, scratch): scratch = the persistent kernel's flag / hand-off buffer (its last 8 words are the watchdog
    words), None when the one-launch-per-timestep kernels ran.  bs_host / offs_host: addresses of the HOST copies of the
    batch-size / offset vectors (the per-step launcher takes them as kernel arguments)."""
    lib = _lib.load()
    dev = gates.device
    st = _lib.stream(dev)
    # row-slot batches (step_masks: ptmi_lstm_forward_persistent_slots): idle rows are not written - they must read as zeros
    c = (torch.zeros if step_masks is not None else torch.empty)((rows, ndir * H), dtype=torch.float32, device=dev)
    rc = -2
    flags = None
    if persistent:
        n = int(lib.ptmi_lstm_scratch_elems(T, ndir, max_batch, H, 0))
        flags = scratch if scratch is not None else torch.empty(n, dtype=torch.int32, device=dev)
        assert flags.numel() >= n and flags.dtype == torch.int32
        if step_masks is not None:
            assert step_masks.dtype == torch.int64 and step_masks.numel() == 3 * T and c0 is None
            rc = _lib.timed('lstm_forward', lib.ptmi_lstm_forward_persistent_slots, gates.data_ptr(), hy.data_ptr(), c.data_ptr(),
                            None, w_hh_pad.data_ptr(), _lib.ptr(w_amax), bs_dev.data_ptr(), offs_dev.data_ptr(), step_masks.data_ptr(),
                            flags.data_ptr(), T, max_batch, rows, H, KP, ndir, int(bool(prefilled and scratch is not None)),
                            _lib.ptr(backward_scratch), st)
        else:
            rc = _lib.timed('lstm_forward', lib.ptmi_lstm_forward_persistent, gates.data_ptr(), hy.data_ptr(), c.data_ptr(),
                            _lib.ptr(c0), w_hh_pad.data_ptr(), _lib.ptr(w_amax), bs_dev.data_ptr(), offs_dev.data_ptr(),
                            flags.data_ptr(), T, max_batch, rows, H, KP, ndir, int(bool(prefilled and scratch is not None)),
                            _lib.ptr(backward_scratch), st)
        if rc not in (0, -2) or (step_masks is not None and rc != 0):
            _lib.check(rc, 'ptmi_lstm_forward_persistent')
    if rc == -2:        # configuration not resident-able: one launch per timestep
        flags = None
        _lib.check(_lib.timed('lstm_forward', lib.ptmi_lstm_forward, gates.data_ptr(), hy.data_ptr(), c.data_ptr(), _lib.ptr(c0),
                              w_hh_pad.data_ptr(), ctypes.c_void_p(bs_host), ctypes.c_void_p(offs_host), T, max_batch, H, KP, ndir,
                              st), 'ptmi_lstm_forward')
    return c, flags


@_register('lstm_recurrence_backward(Tensor gates, Tensor c, Tensor? c0, Tensor dhy, Tensor w_hh_t, Tensor bs_dev, Tensor offs_dev, '
           'int bs_host, int offs_host, int T, int max_batch, int rows, int H, int ndir, bool persistent, Tensor(a!)? scratch=None, '
           'int prefilled=0, Tensor? step_masks=None) -> (Tensor, Tensor?)')
def lstm_recurrence_backward(gates, c, c0, dhy, w_hh_t, bs_dev, offs_dev, bs_host, offs_host, T, max_batch, rows, H, ndir, persistent,
                             scratch=None, prefilled=0, step_masks=None):
    """Returns (dgates, scratch): scratch as above; behind its tile-major copy it carries the bias gradient [ndir * 4H]
    and, for the split kernels, the word with max |dgates| (see ``ops.lstm``)."""
    lib = _lib.load()
    dev = gates.device
    st = _lib.stream(dev)
    dg = torch.empty_like(gates)
    rc = -2
    flags = None
    if persistent:
        n = int(lib.ptmi_lstm_scratch_elems(T, ndir, max_batch, H, 1))
        flags = scratch if scratch is not None else torch.empty(n, dtype=torch.int32, device=dev)
        assert flags.numel() >= n and flags.dtype == torch.int32
        if step_masks is not None:      # row-slot batch
            assert c0 is None
            rc = _lib.timed('lstm_backward', lib.ptmi_lstm_backward_persistent_slots, gates.data_ptr(), c.data_ptr(), dhy.data_ptr(),
                            w_hh_t.data_ptr(), dg.data_ptr(), None, bs_dev.data_ptr(), offs_dev.data_ptr(), step_masks.data_ptr(),
                            flags.data_ptr(), T, max_batch, rows, H, ndir, int(prefilled) if scratch is not None else 0, st)
        else:
            rc = _lib.timed('lstm_backward', lib.ptmi_lstm_backward_persistent, gates.data_ptr(), c.data_ptr(), _lib.ptr(c0),
                            dhy.data_ptr(), w_hh_t.data_ptr(), dg.data_ptr(), bs_dev.data_ptr(), offs_dev.data_ptr(),
                            flags.data_ptr(), T, max_batch, rows, H, ndir, int(prefilled) if scratch is not None else 0, st)
        if rc not in (0, -2) or (step_masks is not None and rc != 0):
            _lib.check(rc, 'ptmi_lstm_backward_persistent')
    if rc == -2:
        flags = None
        dcs = torch.empty((max_batch, ndir, H), dtype=torch.float32, device=dev)
        _lib.check(_lib.timed('lstm_backward', lib.ptmi_lstm_backward, gates.data_ptr(), c.data_ptr(), _lib.ptr(c0), dhy.data_ptr(),
                              w_hh_t.data_ptr(), dg.data_ptr(), dcs.data_ptr(), ctypes.c_void_p(bs_host), ctypes.c_void_p(offs_host),
                              T, max_batch, H, ndir, st), 'ptmi_lstm_backward')
    return dg, flags
