"""(Log-)mel features on the MI355X (reference: ``padertorch/contrib/je/modules/features.py``).

``MelTransform`` keeps the reference's constructor, ``forward`` / ``inverse`` semantics and its
``fbanks`` parameter (``[F, M]``, rows of the filterbank normalised by ``sum + 1e-6``,
``features.py:284-295``); the product ``spectrogram @ fbanks`` followed by ``log(x + eps)``
(``:297-330``) runs as one HIP kernel over the band-compressed filterbank (each triangular filter
touches a few bins only), and :func:`stft_logmel` fuses it behind the STFT so that the spectrum never
reaches HBM (the extractor front-end ``:171-176``).

The filterbank itself is paderbox's ``get_fbanks`` in the reference - a third-party function that is
absent from the reference tree and pinned by none of its tests.  It is restated here from paderbox's
published behaviour (HTK mel scale, ``number_of_filters + 2`` boundaries equally spaced in mel,
unit-peak triangles evaluated at the FFT bin frequencies); parity of THAT matrix to paderbox is
unverified.  ``MelTransform(fbanks=...)`` accepts any ``[F, M]`` matrix instead.
Frequency warping (a random CPU-side augmentation) is out of scope.
"""
from typing import Optional

import numpy as np
import torch
from torch import nn

from .... import _lib
from ....ops._stft import STFT

__all__ = ['MelTransform', 'get_fbanks', 'stft_logmel']


def _hz2mel(f, htk_mel):
    f = np.asarray(f, dtype=np.float64)
    if htk_mel:
        return 2595. * np.log10(1. + f / 700.)
    lin, knee, step = 200. / 3., 1000., np.log(6.4) / 27.
    return np.where(f >= knee, knee / lin + np.log(np.maximum(f, 1e-300) / knee) / step, f / lin)


def _mel2hz(m, htk_mel):
    m = np.asarray(m, dtype=np.float64)
    if htk_mel:
        return 700. * (np.power(10., m / 2595.) - 1.)
    lin, knee, step = 200. / 3., 1000., np.log(6.4) / 27.
    return np.where(m >= knee / lin, knee * np.exp(step * (m - knee / lin)), lin * m)


def get_fbanks(sample_rate: int, stft_size: int, number_of_filters: int, lowest_frequency: float = 0.,
               highest_frequency: Optional[float] = None, htk_mel: bool = True) -> np.ndarray:
    """Unit-peak triangular mel filters ``[number_of_filters, stft_size // 2 + 1]`` (float64)."""
    nyquist = sample_rate / 2
    hi = nyquist if highest_frequency is None else highest_frequency
    lo = 0. if lowest_frequency is None else lowest_frequency
    lo = nyquist + lo if lo < 0 else lo
    hi = nyquist + hi if hi < 0 else hi
    edges = _mel2hz(np.linspace(_hz2mel(lo, htk_mel), _hz2mel(hi, htk_mel), number_of_filters + 2), htk_mel)
    bins = np.arange(stft_size // 2 + 1) * (sample_rate / stft_size)
    left, mid, right = edges[:-2, None], edges[1:-1, None], edges[2:, None]
    rising = (bins[None] - left) / (mid - left)
    falling = (right - bins[None]) / (right - mid)
    return np.clip(np.minimum(rising, falling), 0., None)


class _Bands:
    """Band-compressed copy of an ``[F, M]`` filterbank, cached per device: filter m covers the
    16-byte aligned bin groups ``4 lo[m] .. 4 lo[m] + 8 cnt[m]`` with zero-padded weights at
    ``w[4 off[m]:]`` (the layout ``ptmi_mel_apply`` / ``ptmi_stft_logmel`` read with b128 loads)."""

    def __init__(self, fbanks: np.ndarray):
        F, M = fbanks.shape
        lo, cnt, off, w = [], [], [], []
        for m in range(M):
            nz = np.flatnonzero(fbanks[:, m])
            a, b = (int(nz[0]), int(nz[-1]) + 1) if len(nz) else (0, 1)     # an empty filter keeps one zero group
            a4 = a // 4 * 4
            groups = (b - a4 + 7) // 8
            seg = np.zeros(groups * 8, np.float32)
            seg[a - a4:b - a4] = fbanks[a:b, m]
            lo.append(a4 // 4)
            cnt.append(groups)
            off.append(len(w) // 4)
            w.extend(seg.tolist())
        self.F, self.M, self.nnz = F, M, len(w)
        self.host = dict(lo=np.array(lo, np.int32), cnt=np.array(cnt, np.int32), off=np.array(off, np.int32),
                         w=np.array(w, np.float32))
        self._dev = {}

    def get(self, device):
        key = (device.type, device.index)
        if key not in self._dev:
            self._dev[key] = {k: torch.from_numpy(v).to(device) for k, v in self.host.items()}
        return self._dev[key]


class MelTransform(nn.Module):
    def __init__(
            self,
            sample_rate: int,
            stft_size: int,
            number_of_filters: int,
            lowest_frequency: Optional[float] = 50.,
            highest_frequency: Optional[float] = None,
            htk_mel=True,
            log: bool = True,
            eps=1e-12,
            *,
            warping_fn=None,
            independent_axis=0,
            fbanks=None,
    ):
        """Transforms a linear spectrogram ``[..., F]`` to a (log) mel spectrogram ``[..., M]``.

        Same arguments as the reference (``features.py:214-282``); ``fbanks`` (extension) supplies the
        unnormalised ``[M, F]`` filterbank, e.g. the output of paderbox's ``get_fbanks``.
        """
        super().__init__()
        if warping_fn is not None:
            raise NotImplementedError('frequency warping is a CPU-side augmentation: out of scope')
        self.sample_rate = sample_rate
        self.stft_size = stft_size
        self.number_of_filters = number_of_filters
        self.lowest_frequency = lowest_frequency
        self.highest_frequency = highest_frequency
        self.htk_mel = htk_mel
        self.log = log
        self.eps = eps
        self.warping_fn = None
        self.independent_axis = [independent_axis] if np.isscalar(independent_axis) else independent_axis
        if fbanks is None:
            fbanks = get_fbanks(sample_rate, stft_size, number_of_filters, lowest_frequency, highest_frequency,
                                htk_mel)
        fbanks = np.asarray(fbanks).astype(np.float32)
        assert fbanks.shape == (number_of_filters, stft_size // 2 + 1), fbanks.shape
        fbanks = fbanks / (fbanks.sum(axis=-1, keepdims=True) + 1e-6)           # features.py:292
        self.fbanks = nn.Parameter(torch.from_numpy(np.ascontiguousarray(fbanks.T)), requires_grad=False)
        self._bands = _Bands(np.ascontiguousarray(fbanks.T))

    def forward(self, x, return_maxima=False):
        _lib.require_gpu(x)
        if x.requires_grad:
            raise NotImplementedError('MelTransform is a feature front-end (the reference runs it under '
                                      'no_grad, features.py:156); no backward is provided')
        lib = _lib.load()
        lead, F = x.shape[:-1], x.shape[-1]
        assert F == self._bands.F, (F, self._bands.F)
        spec = x.to(torch.float32).reshape(-1, F).contiguous()
        out = torch.empty((spec.shape[0], self._bands.M), dtype=torch.float32, device=x.device)
        tb = self._bands.get(x.device)
        if spec.shape[0] > 0:           # (an empty tensor has no device pointer)
            _lib.check(_lib.timed(
                'mel_apply', lib.ptmi_mel_apply, spec.data_ptr(), spec.shape[0], F, tb['lo'].data_ptr(),
                tb['cnt'].data_ptr(), tb['off'].data_ptr(), tb['w'].data_ptr(), self._bands.M, self._bands.nnz,
                int(bool(self.log)), float(self.eps), out.data_ptr(), _lib.stream(x.device)), 'ptmi_mel_apply')
        out = out.reshape(*lead, self._bands.M)
        if return_maxima:
            fb = self.fbanks
            maxima = (fb.argmax(-2) + 1) * (fb.sum(-2) > 0) - 1                 # features.py:326-328
            return out, maxima
        return out

    def inverse(self, x):
        """Invert the mel-filterbank transform (``features.py:332-339``; visualisation only)."""
        ifbanks = self.fbanks.T
        ifbanks = ifbanks / (ifbanks.sum(dim=-2, keepdim=True) + 1e-6)
        if self.log:
            x = torch.exp(x)
        x = x @ ifbanks
        return torch.max(x, torch.zeros_like(x))


def stft_logmel(x, stft: STFT, mel: MelTransform, num_samples=None, power: int = 2):
    """Waveforms ``[..., N]`` -> (log-)mel spectrogram ``[..., frames, M]`` in ONE kernel.

    Equals ``mel(abs(stft(x)) ** power)`` - the extractor front-end of the reference
    (``features.py:171-176``: ``mel_transform(torch.sum(x**2, dim=-1))`` on the stacked STFT) - without
    writing the ``[..., frames, F]`` spectrum to HBM.  ``num_samples``: per-row valid samples of a
    padded, ragged batch (frames past a row's end give ``log(eps)``).
    """
    _lib.require_gpu(x)
    if x.requires_grad:
        raise NotImplementedError('stft_logmel is a feature front-end without backward')
    assert power in (1, 2), power
    assert mel.stft_size == stft.size, (mel.stft_size, stft.size)
    lib = _lib.load()
    lead, N = x.shape[:-1], x.shape[-1]
    rows = x.to(torch.float32).reshape(-1, N).contiguous()
    frames = int(lib.ptmi_stft_num_frames(stft._geom, N))
    dev = x.device
    ns = None
    if num_samples is not None:
        ns = torch.as_tensor(num_samples, dtype=torch.int32).reshape(-1).to(dev)
        assert ns.numel() == rows.shape[0], (ns.shape, rows.shape)
    out = torch.empty((rows.shape[0], frames, mel._bands.M), dtype=torch.float32, device=dev)
    tb, mb = stft._tables.get(dev), mel._bands.get(dev)
    rc = _lib.timed(
        'stft_logmel', lib.ptmi_stft_logmel, rows.data_ptr(), rows.shape[0], N, N, _lib.ptr(ns),
        tb['window'].data_ptr(), tb['twiddle'].data_ptr(), stft._geom, frames, mb['lo'].data_ptr(),
        mb['cnt'].data_ptr(), mb['off'].data_ptr(), mb['w'].data_ptr(), mel._bands.M, mel._bands.nnz, power,
        int(bool(mel.log)), float(mel.eps), out.data_ptr(), _lib.stream(dev))
    if rc == -2:
        raise NotImplementedError(f'stft_logmel needs a power-of-two STFT size in 64..2048 (got {stft.size})')
    _lib.check(rc, 'ptmi_stft_logmel')
    return out.reshape(*lead, frames, mel._bands.M)
