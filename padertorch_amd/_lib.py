"""ctypes binding of libptmi.so (C ABI declared in include/ptmi.h).

The product path has NO CPU fallback: if the library is missing, or a tensor is not on an
MI355X device, the ops raise instead of silently computing something else.
"""
import ctypes
import os
import threading
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p, POINTER
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / 'libptmi.so'
if os.environ.get('PTMI_LIB'):            # A/B runs against another build of the same ABI (experiments only)
    LIB_PATH = Path(os.environ['PTMI_LIB'])


class StftGeom(ctypes.Structure):
    """``ptmi_stft_geom`` (include/ptmi.h)."""
    _fields_ = [('size', c_int32), ('shift', c_int32), ('window_length', c_int32),
                ('pad_left', c_int32), ('pad_right', c_int32), ('pad', c_int32)]


class NormGeom(ctypes.Structure):
    """``ptmi_norm_geom`` (include/ptmi.h)."""
    _fields_ = [('rank', c_int32), ('size', c_int64 * 5), ('stat_group_stride', c_int64 * 5),
                ('indep_stride', c_int64 * 5), ('batch_dim', c_int32), ('seq_dim', c_int32)]


_P = c_void_p   # device pointers travel as integers
_G = POINTER(StftGeom)
_I64P = POINTER(c_int64)

#: name -> (restype, argtypes); must list every symbol include/ptmi.h declares
SIGNATURES = {
    'ptmi_version': (c_char_p, []),
    'ptmi_error_string': (c_char_p, [c_int]),
    'ptmi_stft_num_frames': (c_int64, [_G, c_int64]),
    'ptmi_istft_num_samples': (c_int64, [_G, c_int64]),
    'ptmi_stft_forward': (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, _P, _G, c_int64, c_int32,
                                  c_float, _P, _P]),
    'ptmi_istft_forward': (c_int, [_P, c_int64, c_int64, _P, _P, _P, _G, c_int32, c_float, c_int64,
                                   c_int64, c_int64, _P, _P]),
    'ptmi_pit_features': (c_int, [_P, _P, c_int64, c_int32, c_int64, c_int64, _P, _P, _P, _G, c_int64,
                                  _P, _P, _P, _P]),
    'ptmi_pit_features_packed': (c_int, [_P, _P, c_int64, c_int32, c_int64, c_int64, _P, _P, _P, _G, c_int64,
                                         _P, _P, _P, _P, _P, _P, _P]),
    'ptmi_pit_workspace_elems': (c_int64, [c_int64, c_int64, c_int32, c_int32]),
    'ptmi_pit_pairwise_sse': (c_int, [_P, _P, _P, _P, c_int64, c_int64, _I64P, c_int32, c_int32, _P,
                                      _P, _P, _P]),
    'ptmi_pit_assign': (c_int, [_P, c_int64, c_int32, c_int32, c_int32, c_int64, _P, _P, _P, _P, _P]),
    'ptmi_pit_backward': (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int64, _I64P, c_int32, c_int32,
                                  c_int32, _P, _P, _P]),
    'ptmi_dc_workspace_elems': (c_int64, [c_int64, c_int64, c_int32]),
    'ptmi_dc_loss_forward': (c_int, [_P, _P, c_int64, c_int64, _I64P, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P]),
    'ptmi_dc_loss_backward': (c_int, [_P, _P, _P, _P, c_int64, c_int64, _I64P, c_int32, c_int32, c_int32, _P, _P, _P]),
    'ptmi_stft_logmel': (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, _P, _G, c_int64, _P, _P, _P, _P,
                                 c_int32, c_int32, c_int32, c_int32, c_float, _P, _P]),
    'ptmi_mel_apply': (c_int, [_P, c_int64, c_int32, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_float, _P, _P]),
    'ptmi_norm_workspace_elems': (c_int64, [POINTER(NormGeom), c_int32]),
    'ptmi_norm_reduce': (c_int, [c_int32, _P, _P, _P, _P, _P, _P, POINTER(NormGeom), c_int32, _P, _P, _P]),
    'ptmi_norm_elementwise': (c_int, [c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, POINTER(NormGeom), c_int32,
                                      c_int32, _P, _P]),
    'ptmi_td_stats_elems': (c_int64, [c_int32]),
    'ptmi_td_workspace_elems': (c_int64, [c_int64, c_int32, c_int64]),
    'ptmi_td_pair_stats': (c_int, [_P, _P, _P, c_int64, c_int32, c_int64, _I64P, _P, _P, _P]),
    'ptmi_td_lincomb': (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int64, _I64P, _P, _P]),
    'ptmi_lstm_forward': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    'ptmi_lstm_backward': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    'ptmi_unit_norm_forward': (c_int, [_P, _P, _P, c_int64, c_int32, c_int32, c_float, c_void_p]),
    'ptmi_unit_norm_backward': (c_int, [_P, _P, _P, _P, c_int64, c_int32, c_int32, c_float, c_void_p]),
    'ptmi_lstm_flags_elems': (c_int64, [c_int32, c_int32, c_int32]),
    'ptmi_lstm_scratch_elems': (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    'ptmi_lstm_weight_prep': (c_int, [_P, _P, _P, _P, c_int32, c_int32, c_int32, _P, c_int32, _P, _P, c_int32, _P, _P, _P]),
    'ptmi_lstm_set_error_sink': (c_int, [_P]),
    'ptmi_lstm_split_enabled': (c_int, []),
    'ptmi_lstm_handoff_cols': (c_int32, [c_int32, c_int32]),
    'ptmi_lstm_forward_persistent': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, c_int32, c_int32,
                                             c_int32, c_int32, _P, _P]),
    'ptmi_lstm_forward_fills': (c_int, [c_int32, c_int32, c_int32, c_int32]),
    'ptmi_lstm_backward_persistent': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, c_int32,
                                              c_int32, c_int32, _P]),
    'ptmi_lstm_backward_persistent_range': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, c_int32,
                                                    c_int32, c_int32, c_int32, c_int32, _P]),
    'ptmi_lstm_forward_persistent_slots': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, c_int32, c_int32,
                                                   c_int32, c_int32, _P, _P]),
    'ptmi_lstm_backward_persistent_slots': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, c_int32, c_int32,
                                                    c_int32, _P]),
    'ptmi_lstm_backward_persistent_states': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, c_int32,
                                                     c_int32, c_int32, _P]),
    'ptmi_lstm_backward_planes_ok': (c_int32, [c_int32, c_int32, c_int32, c_int64, c_int32]),
    'ptmi_lstm_backward_persistent_planes': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, c_int32,
                                                     c_int32, c_int32, c_int32, c_int32, _P]),
    'ptmi_lstm_scratch_prefill': (c_int, [_P, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    'ptmi_absmax': (c_int, [_P, c_int64, c_int64, c_int64, _P, _P]),
    'ptmi_absmax_accumulate': (c_int, [_P, c_int64, c_int64, c_int64, _P, _P]),
    'ptmi_planes_elems': (c_int64, [c_int64, c_int64]),
    'ptmi_pack_planes_t': (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, _P]),
    'ptmi_gemm_planes_workspace_elems': (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    'ptmi_pack_planes_n': (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, _P]),
    'ptmi_gemm_planes': (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    'ptmi_gemm_planes_relu': (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P, c_int32, _P]),
    'ptmi_relu_backward_absmax': (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int64, c_int64, _P, c_int32, _P]),
    'ptmi_pack_planes_t_bf16': (c_int, [_P, c_int64, c_int64, c_int64, _P, _P]),
    'ptmi_pack_planes_n_bf16': (c_int, [_P, c_int64, c_int64, c_int64, _P, _P]),
    'ptmi_pack_planes_into': (c_int, [_P, c_int64, c_int64, c_int64, c_int32, c_int32, _P, _P, c_int64, c_int64, c_int64, _P]),
    'ptmi_gemm_planes_bf16': (c_int, [_P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    'ptmi_gemm_planes_bf16_two': (c_int, [_P, _P, _P, _P, c_int32, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    'ptmi_gemm_planes_plan': (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    'ptmi_comm_rccl_version': (c_int32, []),
    'ptmi_comm_unique_id': (c_int, [_P]),
    'ptmi_comm_create': (c_int, [_P, c_int32, c_int32, _P]),
    'ptmi_allreduce_sum': (c_int, [_P, _P, c_int64, _P]),
    'ptmi_comm_destroy': (c_int, [_P]),
    'ptmi_lstm_bias_grad_add': (c_int, [_P, c_int32, c_int32, _P, _P, _P]),
    'ptmi_grad_norm_workspace_elems': (c_int64, []),
    'ptmi_grad_norm': (c_int32, [_P, c_int64, _P, _P, _P]),
    'ptmi_adam_flat': (c_int32, [_P, _P, _P, _P, c_int32, c_int64, _P, c_float, _P, _P, _P, _P, c_double, c_double, c_double,
                                 c_double, c_double, _P, c_int32, _P]),
}

_lib = None


def load():
    """Load libptmi.so (once).  Raises if it has not been built - there is no fallback."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f'{LIB_PATH} is missing: build the HIP kernels first '
                f'(python -m padertorch_amd.build, or __graft_entry__.build()). '
                f'padertorch_amd has no CPU fallback.')
        lib = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


_hooks = None


def test_hooks():
    """``libptmi_testhooks.so`` (``csrc/testhooks``: ``ptmi_test_occupy``) - kernels only tests and measurement brackets launch; not part
    of the hot path's library."""
    global _hooks
    if _hooks is None:
        path = _PKG / 'libptmi_testhooks.so'
        if not path.exists():
            raise ImportError(f'{path} is missing: python -m padertorch_amd.build')
        lib = ctypes.CDLL(str(path))
        lib.ptmi_test_occupy.restype = c_int
        lib.ptmi_test_occupy.argtypes = [c_int32, c_int32, c_int32, c_int64, _P]
        _hooks = lib
    return _hooks


def select_gemm_tile(tile):
    """Pin the planes GEMM's workgroup tile for the calls that follow (0..5; -1 / None: the cost model) - the ``PTMI_GEMM_TILE`` variable
    ``csrc/gemm_planes.hip`` reads at every call: tests and sweeps only."""
    if tile is None or tile < 0:
        os.environ.pop('PTMI_GEMM_TILE', None)
    else:
        assert 0 <= tile <= 5, tile
        os.environ['PTMI_GEMM_TILE'] = str(int(tile))
    return 0


def check(rc, what=''):
    if rc != 0:
        msg = load().ptmi_error_string(int(rc)).decode()
        raise RuntimeError(f'{what}: {msg} (code {rc})')


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                'padertorch_amd ops run on an MI355X (HIP) device only and have no CPU fallback; '
                f'got a tensor on {t.device}. Move the example to the GPU first.')


def stream(device):
    return torch.cuda.current_stream(device).cuda_stream


#: when a list, ``timed`` brackets every kernel launch with HIP events recorded on the launch stream
#: and appends ``(name, start_event, end_event)`` (bench.py's roofline figure); None = off
KERNEL_TIMERS = None


def timed(name, fn, *args):
    """Call a C-ABI launcher; optionally bracket it with events on torch's current stream."""
    if KERNEL_TIMERS is None:
        return fn(*args)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    KERNEL_TIMERS.append((name, e0, e1))
    return rc


class _Staging:
    """Pinned staging memory for small host -> device copies: two halves used in turn.  One event per half (recorded when the half is
    left) says when the copies out of it have run; a half is written again only behind its event - long fired by then.  (A fresh
    ``pin_memory()`` per copy works too, but its allocator has to make a new pinned block whenever the previous step's block is still in
    flight: 70 ms of hipHostMalloc now and then for the 0.5 MB index tables of a 64 x 6 s batch.)"""

    def __init__(self, half_bytes=1 << 24):
        self.half = half_bytes
        self.buf = torch.empty(2 * half_bytes, dtype=torch.uint8, pin_memory=True)
        self.np = self.buf.numpy()            # the same memory for numpy: a plain memcpy fills it (torch's CPU copy_ of > 32 k elements
        #                                       goes through the intra-op thread pool: 0.4 ms and more on a loaded many-core host)
        self.which, self.off = 0, 0
        self.events = [[], []]           # per half: events behind the copies out of it (one per stream that copied)
        self.streams = [{}, {}]          # per half: the streams that have copied out of it since it was entered
        self.lock = threading.Lock()

    def put(self, t, device):
        n = t.numel() * t.element_size()
        if n > self.half // 4:
            return t.pin_memory().to(device, non_blocking=True)
        with self.lock:
            if self.off + n > self.half:                       # leave this half: an event behind everything copied out of it
                for stream in self.streams[self.which].values():
                    ev = torch.cuda.Event()
                    ev.record(stream)
                    self.events[self.which].append(ev)
                self.streams[self.which] = {}
                self.which, self.off = self.which ^ 1, 0
                for ev in self.events[self.which]:
                    ev.synchronize()
                self.events[self.which] = []
            start = self.which * self.half + self.off
            self.off += (n + 63) // 64 * 64
            stage = self.buf[start:start + n].view(t.dtype).view(t.shape)
            self.np[start:start + n] = t.numpy().reshape(-1).view('uint8')
            cur = torch.cuda.current_stream(device)
            self.streams[self.which][cur.cuda_stream] = cur
            return stage.to(device, non_blocking=True)


_STAGING = {}


def host_to_device(values, dtype, device):
    """A small host list / ndarray as a device tensor WITHOUT a host synchronisation: pinned staging memory + non-blocking copy.
    (``torch.tensor(values, device='cuda')`` copies from pageable memory, which blocks the host until the stream has drained: with
    one such call per step - the lengths of a ragged batch - the host never runs ahead of the GPU and every launch gap of the step
    becomes visible: 12.7 instead of 10.0 ms per step for 32 examples of 3-6 s)"""
    t = torch.as_tensor(values, dtype=dtype)
    device = torch.device(device)
    if device.type != 'cuda' or t.numel() == 0:
        return t.to(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _STAGING.get(idx)
    if st is None:
        st = _STAGING[idx] = _Staging()
    return st.put(t.contiguous(), torch.device('cuda', idx))


def strides4(*vals):
    return (c_int64 * 4)(*vals)


def strides6(*vals):
    return (c_int64 * 6)(*vals)


def strides8(*vals):
    return (c_int64 * 8)(*vals)


#: PTMI_STRICT=1 (or ``padertorch_amd._lib.STRICT = True``): a request that would leave the hand-written HIP path - an LSTM the
#: recurrence kernels do not cover, a dense layer handed to the BLAS library - raises instead of warning.  The GPU test suite runs strict.
STRICT = bool(int(os.environ.get('PTMI_STRICT', '0') or 0))
_WARNED = set()


def leaving_native_path(what, reason):
    """``what`` is about to run on a library kernel (MIOpen / rocBLAS through torch) instead of this package's HIP kernels because
    of ``reason``: said once per (what, reason) as a ``RuntimeWarning``, or raised under :data:`STRICT` - never silently."""
    msg = f'padertorch_amd: {what} runs on the torch / library path, not on the HIP kernels: {reason}'
    if STRICT:
        raise RuntimeError(msg + ' (PTMI_STRICT)')
    if (what, reason) not in _WARNED:
        _WARNED.add((what, reason))
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)
