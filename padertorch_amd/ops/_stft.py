"""``padertorch_amd.ops.STFT``: drop-in for ``padertorch.ops.STFT`` on MI355X.

Same constructor / ``__call__`` / ``inverse`` / helper signatures and assertion behaviour as
``padertorch/ops/_stft.py:46-307``; the arithmetic runs in hand-written HIP kernels
(``csrc/stft.hip``) through the C ABI of ``include/ptmi.h``:

* ``__call__``  -> ``ptmi_stft_forward``  (reference: F.pad + F.pad + kernel.to(x) + F.conv1d with a
  dense ``[2F,1,L]`` DFT matrix, ``_stft.py:131-174``)
* ``inverse``   -> ``ptmi_istft_forward`` (reference: two ``conv_transpose1d``, ``_stft.py:226-262``)
* both are differentiable; each op's backward is the other kernel (adjoint), see DESIGN.md.

Like the reference it is a plain class (no parameters, not in ``state_dict``).  The window /
synthesis-window / twiddle tables are built once on the host in float64 (window semantics of
paderbox ``_get_window`` / ``_biorthogonal_window_fastest``: periodic unless
``symmetric_window``; synthesis window ``w[n] / sum_m w[n+m*shift]**2``) and cached per device.
"""
import typing
from math import ceil

import numpy as np
import torch

from .. import _lib
from . import library  # noqa: F401  (registers torch.ops.ptmi.*)

__all__ = ['STFT']


def _get_window(window, symmetric_window, window_length):
    """paderbox ``_get_window`` semantics (reference call site ``_stft.py:91-95``)."""
    import scipy.signal
    fn = window if callable(window) else getattr(scipy.signal.windows, window)
    if symmetric_window:
        return np.asarray(fn(window_length), dtype=np.float64)
    return np.asarray(fn(window_length + 1)[:-1], dtype=np.float64)


def _biorthogonal_window(window, shift):
    """paderbox ``_biorthogonal_window_fastest`` semantics (reference call site ``_stft.py:27-28``)."""
    w = np.asarray(window, dtype=np.float64)
    sq = w * w
    denom = np.empty_like(w)
    for r in range(min(shift, len(w))):
        denom[r::shift] = sq[r::shift].sum()
    return w / denom


def _fading_pads(window_length, shift, fading):
    """``_stft.py:137-146``."""
    if fading in (None, False):
        return 0, 0
    if fading == 'half':
        return (window_length - shift) // 2, ceil((window_length - shift) / 2)
    return window_length - shift, window_length - shift


class _Tables:
    """Per-device float32 tables derived from float64 host arrays."""

    def __init__(self, window, shift, size):
        self.host = dict(
            window=window,
            syn=_biorthogonal_window(window, shift) / size,          # _stft.py:27-28
            # adjoint windows (DESIGN.md "adjoints"): d stft -> istft with w/2; d istft -> stft with 2*syn
            window_adj=window / 2,
            syn_adj=2 * _biorthogonal_window(window, shift) / size,
        )
        j = np.arange(size // 2 + 1)
        ang = 2 * np.pi * j / size
        self.host['twiddle'] = np.stack([np.cos(ang), -np.sin(ang)], axis=-1).reshape(-1)
        self._dev = {}

    def get(self, device):
        key = (device.type, device.index)
        if key not in self._dev:
            self._dev[key] = {k: torch.as_tensor(v, dtype=torch.float32).to(device)
                              for k, v in self.host.items()}
        return self._dev[key]


def _to_f32(x):
    if x.dtype != torch.float32:
        x = x.to(torch.float32)       # arithmetic is fp32 (documented deviation for f64 inputs)
    return x.contiguous()


def _geom_list(st):
    g = st._geom
    return [g.size, g.shift, g.window_length, g.pad_left, g.pad_right, g.pad]


class _StftFn(torch.autograd.Function):
    """[rows, T] float32 -> [rows, frames, F, 2] (interleaved) or [rows, frames, 2F] (concat); kernels:
    ``torch.ops.ptmi.stft_forward`` and, as its adjoint, ``torch.ops.ptmi.istft_forward``."""

    @staticmethod
    def forward(ctx, x, st, layout, row_samples):
        _lib.require_gpu(x)
        tb = st._tables.get(x.device)
        T = x.shape[1]
        out = torch.ops.ptmi.stft_forward(x, row_samples, tb['window'], tb['twiddle'], _geom_list(st), st._frames_for(T), layout,
                                          1.0, 'stft_forward')
        ctx.st, ctx.layout, ctx.T = st, layout, T
        return out

    @staticmethod
    def backward(ctx, g):
        st, T = ctx.st, ctx.T
        tb = st._tables.get(g.device)
        # adjoint of (frame, window, one-sided DFT) = hermitian inverse with doubled DC/Nyquist,
        # window w/2, overlap-add, cut the fading pad and anything right of the row.
        dx = torch.ops.ptmi.istft_forward(g.contiguous(), tb['window_adj'], tb['twiddle'], _geom_list(st), ctx.layout, 2.0,
                                          st._geom.pad_left, T, '')
        return dx, None, None, None


class _IstftFn(torch.autograd.Function):
    """[rows, frames, F, 2] or [rows, frames, 2F] float32 -> [rows, samples]."""

    @staticmethod
    def forward(ctx, spec, st, layout):
        _lib.require_gpu(spec)
        tb = st._tables.get(spec.device)
        frames = spec.shape[1]
        n = int(_lib.load().ptmi_istft_num_samples(st._geom, frames))
        out = torch.ops.ptmi.istft_forward(spec, tb['syn'], tb['twiddle'], _geom_list(st), layout, 1.0, st._geom.pad_left, n,
                                           'istft_forward')
        ctx.st, ctx.layout, ctx.frames = st, layout, frames
        return out

    @staticmethod
    def backward(ctx, g):
        st, frames = ctx.st, ctx.frames
        tb = st._tables.get(g.device)
        # adjoint of (hermitian inverse, window, overlap-add, cut) = forward STFT of the gradient
        # with window 2*syn and halved, real-only DC/Nyquist bins.
        ds = torch.ops.ptmi.stft_forward(g.contiguous(), None, tb['syn_adj'], tb['twiddle'], _geom_list(st), frames, ctx.layout,
                                         0.5, '')
        return ds, None, None


class STFT:
    def __init__(
            self,
            size: int = 1024,
            shift: int = 256,
            *,
            window: typing.Union[str, typing.Callable] = 'blackman',
            window_length: int = None,
            fading: typing.Optional[typing.Union[bool, str]] = 'full',
            pad: bool = True,
            symmetric_window: bool = False,
            complex_representation: str = 'complex'
    ):
        """Arguments as ``padertorch.ops.STFT`` (``_stft.py:47-58``)."""
        self.possible_out_types = ['concat', 'stacked', 'complex']
        assert complex_representation in self.possible_out_types, (
            f'Please choose one of the predefined output_types'
            f' {self.possible_out_types}, not {complex_representation}'
        )
        self.complex_representation = complex_representation
        assert size % 2 == 0, 'At the moment we only support even FFT sizes'
        self.size = size
        self.shift = shift
        self.window_length = window_length if window_length is not None else size
        assert self.window_length <= size, (self.window_length, size)
        assert fading in [None, True, False, 'full', 'half'], fading
        self._fading = fading
        self._pad = pad
        self.window = _get_window(window, symmetric_window, self.window_length)
        self._tables = _Tables(self.window, shift, size)
        self._update_geom()

    # ``fading`` / ``pad`` may be reassigned after construction (tests/test_ops/test_stft.py:45-58)
    @property
    def fading(self):
        return self._fading

    @fading.setter
    def fading(self, value):
        assert value in [None, True, False, 'full', 'half'], value
        self._fading = value
        self._update_geom()

    @property
    def pad(self):
        return self._pad

    @pad.setter
    def pad(self, value):
        self._pad = value
        self._update_geom()

    def _update_geom(self):
        left, right = _fading_pads(self.window_length, self.shift, self._fading)
        self._geom = _lib.StftGeom(self.size, self.shift, self.window_length, left, right,
                                   1 if self._pad else 0)

    def _frames_for(self, samples):
        """Exact frame count of the strided conv in ``_stft.py:158`` (integer arithmetic)."""
        n = int(_lib.load().ptmi_stft_num_frames(self._geom, int(samples)))
        if n <= 0:
            raise RuntimeError(
                f'STFT: input of {samples} samples is shorter than the window '
                f'({self.window_length}) and pad=False')
        return n

    def __call__(self, inputs, num_samples=None):
        """``inputs``: ``[..., T]`` -> ``[..., frames, F]`` complex64 (or concat / stacked).

        ``num_samples`` (extension, optional int32 tensor with one entry per flattened row): rows
        are zero-padded to ``T``; frames past a row's own count are zero.
        """
        org_shape = inputs.shape
        x = _to_f32(inputs.reshape(-1, org_shape[-1]))
        layout = 1 if self.complex_representation == 'concat' else 0
        out = _StftFn.apply(x, self, layout, num_samples)
        out = out.reshape(*org_shape[:-1], *out.shape[1:])
        if self.complex_representation == 'complex':
            out = torch.view_as_complex(out)
        if inputs.dtype == torch.float64:
            out = out.to(torch.complex128 if out.is_complex() else torch.float64)
        return out

    def inverse(self, stft_signal):
        """``[..., frames, F]`` complex / ``[..., frames, 2F]`` / ``[..., frames, F, 2]`` -> ``[..., T]``."""
        if self.complex_representation == 'complex':
            assert stft_signal.is_complex(), stft_signal.dtype
            dbl = stft_signal.dtype == torch.complex128
            spec = torch.view_as_real(stft_signal.to(torch.complex64).contiguous())
            lead, layout = stft_signal.shape[:-2], 0
        elif self.complex_representation == 'stacked':
            dbl = stft_signal.dtype == torch.float64
            spec, lead, layout = _to_f32(stft_signal), stft_signal.shape[:-3], 0
        elif self.complex_representation == 'concat':
            dbl = stft_signal.dtype == torch.float64
            spec, lead, layout = _to_f32(stft_signal), stft_signal.shape[:-2], 1
        else:
            raise ValueError(
                f'Please choose one of the predefined output_types'
                f'{self.possible_out_types} not {self.complex_representation}')
        F = self.size // 2 + 1
        frames = spec.shape[len(lead)]
        spec = spec.reshape(-1, frames, *((F, 2) if layout == 0 else (2 * F,)))
        out = _IstftFn.apply(spec, self, layout)
        out = out.reshape(*lead, out.shape[-1])
        return out.to(torch.float64) if dbl else out

    def samples_to_frames(self, samples):
        """paderbox ``_samples_to_stft_frames(samples, window_length, shift, pad, fading)``."""
        if self._fading not in (None, False):
            samples = samples + (1 + (self._fading != 'half')) * (self.window_length - self.shift)
        frames = (samples - self.window_length + self.shift) / self.shift
        if isinstance(frames, np.ndarray):
            return (np.ceil(frames) if self._pad else np.floor(frames)).astype(np.int64)
        return ceil(frames) if self._pad else int(np.floor(frames))

    def sample_index_to_frame_index(self, sample_index):
        """Index of the frame that represents ``sample_index`` best: the last frame whose window centre
        (``start + window_length // 2``) is not behind the sample, at least 0; with fading the frame grid starts
        ``pad_left`` samples before the signal.

        The reference defers to ``paderbox.transform.module_stft.sample_index_to_stft_frame_index``
        (``_stft.py:281-293``), which is third-party, absent from the reference tree and exercised by no reference
        test: this restates its published behaviour (``[f(i, 8, 1, fading=None) for i in range(12)] ==
        [0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7]``; fading adds the ``ceil(pad / shift)`` frames in front) and its
        parity to paderbox is UNPINNED, like the mel filterbank."""
        lead = 0
        if self._fading not in (None, False):
            pad_left = self.window_length - self.shift
            if self._fading == 'half':
                pad_left //= 2
            lead = -(-pad_left // self.shift)
        centre = self.window_length // 2
        if isinstance(sample_index, np.ndarray):
            return np.maximum((sample_index - centre) // self.shift, 0).astype(np.int64) + lead
        return max((int(sample_index) - centre) // self.shift, 0) + lead

    def frames_to_samples(self, frames):
        """paderbox ``_stft_frames_to_samples(frames, window_length, shift, fading)``."""
        samples = frames * self.shift + self.window_length - self.shift
        if self._fading not in (None, False):
            samples = samples - (1 + (self._fading != 'half')) * (self.window_length - self.shift)
        return samples
