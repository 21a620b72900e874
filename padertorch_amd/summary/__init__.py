"""Image helpers called from the PIT review (``padertorch/summary/tbx_utils.py:61-157``).

Only evaluated when ``model.create_snapshot`` is set, so the D2H sync they imply does not sit on
the training step (SURVEY.md section 3.2 / hard parts).  Grayscale output ``(1, F, T)`` uint8.
"""
import numpy as np
import torch

__all__ = ['mask_to_image', 'stft_to_image']


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def mask_to_image(mask, origin='lower'):
    """Clip to [0, 1] -> uint8 image, features on the y axis (``tbx_utils.py:61-104``)."""
    image = np.clip(_np(mask) * 255, 0, 255).astype(np.uint8).T
    if origin == 'lower':
        image = image[::-1]
    return image[None]


def stft_to_image(signal, origin='lower', visible_dB=50):
    """Power spectrogram in dB relative to its maximum, ``visible_dB`` mapped to 0..255."""
    s = _np(signal)
    power = np.abs(s) ** 2
    floor = 10 ** (-visible_dB / 10)
    p = np.maximum(power / max(power.max(), np.finfo(np.float64).tiny), floor)
    image = ((10 * np.log10(p) + visible_dB) / visible_dB * 255).astype(np.uint8).T
    if origin == 'lower':
        image = image[::-1]
    return image[None]
