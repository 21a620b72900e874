"""On-device PIT feature front-end.

Replaces the numpy data-pipeline transform ``pre_batch_transform``
(``padertorch/contrib/examples/source_separation/pit/data.py:49-77``: ``stft(s,512,128)``,
``stft(y,512,128)``, ``|.|`` and ``cos(angle(Y) - angle(X))``) by one fused HIP kernel
(``ptmi_pit_features``) that turns the raw waveforms of a whole batch into the model's inputs
directly in HBM - the H2D copy shrinks from (1+2K) feature maps to (1+K) waveforms
 and the STFTs never leave the device.
"""
import torch

from .. import _lib
from . import library  # noqa: F401  (registers torch.ops.ptmi.*)
from ._stft import STFT
from .sequence.pack_module import PaddedList

__all__ = ['pit_features']

_default_stft = None


def _pad_rows(rows, dim_last_pad_to):
    out = rows[0].new_zeros((len(rows),) + tuple(rows[0].shape[:-1]) + (dim_last_pad_to,))
    for b, r in enumerate(rows):
        out[b, ..., :r.shape[-1]] = r
    return out


def pit_features(y, s=None, num_samples=None, stft: STFT = None):
    """Waveforms -> ``dict(Y_abs, X_abs, cos_phase_difference, num_frames)`` (model batch contract).

    Args:
        y: mixtures: list of ``(N_b,)`` tensors (sorted by descending length) or a padded ``[B, N]``
        s: sources: list of ``(K, N_b)`` tensors or a padded ``[B, K, N]`` (None: only ``Y_abs``)
        num_samples: list of ints when ``y`` / ``s`` are padded and ragged
        stft: an :class:`STFT` (default ``STFT(512, 128)`` = paderbox defaults used by the example)
    Every entry of the result is a :class:`PaddedList` (list of per-example views, e.g.
    ``Y_abs[b]: (T_b, F)``, ``X_abs[b]: (T_b, K, F)``).
    """
    global _default_stft
    if stft is None:
        if _default_stft is None:
            _default_stft = STFT(512, 128)
        stft = _default_stft
    if isinstance(y, (list, tuple)):
        num_samples = [int(t.shape[-1]) for t in y]
        y = _pad_rows(list(y), max(num_samples))
        if s is not None:
            s = _pad_rows(list(s), max(num_samples))
    _lib.require_gpu(y, s)
    assert y.dim() == 2 and y.dtype == torch.float32, (y.shape, y.dtype)
    y = y.contiguous()
    B, N = y.shape
    K = 0
    if s is not None:
        assert s.dim() == 3 and s.shape[0] == B and s.shape[2] == N and s.dtype == torch.float32
        s = s.contiguous()
        K = s.shape[1]
    if num_samples is None:
        num_samples = [N] * B
    lib = _lib.load()
    per_len = {n: int(lib.ptmi_stft_num_frames(stft._geom, n)) for n in set(num_samples)}
    frames = [per_len[n] for n in num_samples]
    T = max(frames)
    F = stft.size // 2 + 1
    dev = y.device
    ragged = any(n != N for n in num_samples)
    ns_dev = torch.tensor(num_samples, dtype=torch.int32, device=dev) if ragged else None
    tb = stft._tables.get(dev)
    g = stft._geom
    Y_abs, X_abs, cos_pd = torch.ops.ptmi.pit_features(
        y, s, ns_dev, tb['window'], tb['twiddle'], [g.size, g.shift, g.window_length, g.pad_left, g.pad_right, g.pad], T)
    fl = torch.tensor(frames, dtype=torch.int32, device=dev) if ragged else None
    out = dict(Y_abs=PaddedList(Y_abs, frames, True, fl), num_frames=frames)
    if K:
        out['X_abs'] = PaddedList(X_abs, frames, True, fl)
        out['cos_phase_difference'] = PaddedList(cos_pd, frames, True, fl)
    return out
