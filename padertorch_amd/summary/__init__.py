"""Image helpers of the review step, same call signatures as the reference
(``padertorch/summary/tbx_utils.py:61-157,219-271``):

    mask_to_image(mask, batch_first=False, color=None, origin='lower')
    stft_to_image(signal, batch_first=False, color='viridis', origin='lower', visible_dB=50)
    spectrogram_to_image(signal, batch_first=False, color='viridis', origin='lower', log=True, visible_dB=50)

All three return ``(channels, features, frames)`` arrays for tensorboard: one uint8 channel when ``color`` is
``None``, the RGBA floats of the matplotlib colour map otherwise (grayscale with a warning when matplotlib is
missing).  They copy their input to the host, so the models only call them when ``create_snapshot`` is set.
"""
import warnings

import numpy as np
import torch

__all__ = ['mask_to_image', 'stft_to_image', 'spectrogram_to_image']

_CMAPS = {}


def _host(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def _frames_by_features(a, batch_first):
    """2-D ``(frames, features)`` view: the first example of a 3-D input (batch axis 0 or 1)."""
    if a.ndim == 2:
        return a
    if a.ndim != 3:
        raise ValueError(f'Either the signal has ndim 2 or 3', a.shape)
    if batch_first is None:
        raise ValueError(f'The array still has a batch axis but batch_first is None. Shape: {a.shape}')
    return a[0] if batch_first else a[:, 0]


def _to_image(levels, origin, color):
    """``(frames, features)`` uint8 levels -> ``(channels, features, frames)``."""
    assert origin in ('upper', 'lower'), origin
    img = levels.T
    if origin == 'lower':
        img = img[::-1]
    if color is None:
        return img[None]
    name = 'viridis' if color is True else color
    if name not in _CMAPS:
        try:
            import matplotlib.pyplot as plt
            _CMAPS[name] = plt.get_cmap(name)
        except ImportError:
            warnings.warn('Since matplotlib is not installed, all images are switched to grey scale')
            _CMAPS[name] = None
    cmap = _CMAPS[name]
    if cmap is None:
        return img[None]
    return np.moveaxis(cmap(img), -1, 0)


def mask_to_image(mask, batch_first=False, color=None, origin='lower'):
    """Mask values in [0, 1] (clipped, with a warning, when they are not) as image levels 0..255."""
    m = _host(mask)
    outside = int(np.sum((m < 0) | (m > 1)))
    if outside:
        warnings.warn(f'Mask value passed to mask_to_image out of range ([0, 1])! {outside} values are clipped!')
    levels = np.clip(m * 255, 0, 255).astype(np.uint8)
    return _to_image(_frames_by_features(levels, batch_first), origin, color)


def spectrogram_to_image(signal, batch_first=False, color='viridis', origin='lower', log=True, visible_dB=50):
    """Power spectrogram, normalised to its maximum (taken over the whole input, batch included); with ``log``
    the top ``visible_dB`` decibels are spread over the levels 0..255."""
    p = _host(signal)
    p = p / (np.max(np.abs(p)) + np.finfo(p.dtype).tiny)
    p = _frames_by_features(p, batch_first)
    if log:
        p = np.maximum(p, 10 ** (-visible_dB / 10))
        p = (10 / visible_dB) * np.log10(p) + 1
    return _to_image((p * 255).astype(np.uint8), origin, color)


def stft_to_image(signal, batch_first=False, color='viridis', origin='lower', visible_dB=50):
    """(Complex or magnitude) STFT -> image of its power in dB."""
    s = _host(signal)
    return spectrogram_to_image(s.real ** 2 + s.imag ** 2, batch_first=batch_first, color=color, origin=origin,
                                visible_dB=visible_dB)
