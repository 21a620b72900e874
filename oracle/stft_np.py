"""numpy (float64) restatement of the reference STFT / iSTFT.  ORACLE - test infrastructure only.

Follows
  * ``padertorch/ops/_stft.py:11-23``   (``get_stft_kernel``: DFT*window matrix)
  * ``padertorch/ops/_stft.py:26-43``   (``get_istft_kernel``)
  * ``padertorch/ops/_stft.py:103-174`` (``STFT.__call__``: fading pad, frame pad, strided conv)
  * ``padertorch/ops/_stft.py:176-263`` (``STFT.inverse``: hermitian extension, conv-transpose, cut)
  * ``padertorch/ops/_stft.py:265-307`` (frame/sample count helpers -> paderbox)
and the third-party helpers it calls in ``paderbox.transform.module_stft``
(``_get_window``, ``_biorthogonal_window_fastest``, ``_samples_to_stft_frames``,
``_stft_frames_to_samples``, ``stft``, ``istft``).  paderbox is NOT in the
reference tree nor in this image; those helpers are restated from their
published behaviour and pinned by the reference's own known answers:
  * periodic hann(4) = [0,.5,1,.5] and the 5x3 matrix in
    ``padertorch/contrib/cb/transform.py:219-232``
  * frame counts in ``tests/test_ops/test_stft.py:44-70,139-165``
  * perfect reconstruction ``tests/test_ops/test_stft.py:36-42`` which fixes the
    synthesis window to ``w[n] / sum_m w[n + m*shift]**2``.
``sample_index_to_stft_frame_index`` has no pin in the reference tree
("parity unpinned").
"""
from math import ceil, floor

import numpy as np
import scipy.signal


# --------------------------------------------------------------------------- window helpers
def get_window(window, symmetric_window=False, window_length=None):
    """paderbox ``_get_window`` (call site ``padertorch/ops/_stft.py:91-95``).

    ``symmetric_window=False`` -> periodic window ``win(L + 1)[:-1]``.
    """
    if callable(window):
        fn = window
    else:
        fn = getattr(scipy.signal.windows, window)
    if symmetric_window:
        return np.asarray(fn(window_length), dtype=np.float64)
    return np.asarray(fn(window_length + 1)[:-1], dtype=np.float64)


def biorthogonal_window(analysis_window, shift):
    """paderbox ``_biorthogonal_window_fastest`` (call site ``_stft.py:27-28``).

    ws[n] = w[n] / sum_{m == n (mod shift)} w[m]**2 : the unique synthesis window for which
    overlap-add of ``frame * ws`` reconstructs the signal.
    """
    w = np.asarray(analysis_window, dtype=np.float64)
    L = len(w)
    denom = np.zeros(L)
    sq = w ** 2
    for r in range(min(shift, L)):
        denom[r::shift] = sq[r::shift].sum()
    return w / denom


def samples_to_frames(samples, window_length, shift, pad=True, fading=None):
    """paderbox ``_samples_to_stft_frames`` (call site ``_stft.py:276-279``)."""
    if fading not in (None, False):
        samples = samples + (1 + (fading != 'half')) * (window_length - shift)
    frames = (samples - window_length + shift) / shift
    if isinstance(frames, np.ndarray):
        return (np.ceil(frames) if pad else np.floor(frames)).astype(np.int64)
    return ceil(frames) if pad else floor(frames)


def frames_to_samples(frames, window_length, shift, fading=None):
    """paderbox ``_stft_frames_to_samples`` (call site ``_stft.py:305-307``)."""
    samples = frames * shift + window_length - shift
    if fading not in (None, False):
        samples = samples - (1 + (fading != 'half')) * (window_length - shift)
    return samples


def fading_pad_width(window_length, shift, fading):
    """``_stft.py:137-146``: (left, right) zero padding for the fade-in/out."""
    if fading in (None, False):
        return 0, 0
    if fading == 'half':
        return (window_length - shift) // 2, ceil((window_length - shift) / 2)
    return window_length - shift, window_length - shift


def padded_length(num_samples, window_length, shift, fading, pad):
    """Length of the virtual zero-padded signal the frames are cut from (``_stft.py:137-154``)."""
    left, right = fading_pad_width(window_length, shift, fading)
    T = num_samples + left + right
    if pad:
        if T < window_length:
            T = window_length
        elif shift != 1 and (T + shift - window_length) % shift != 0:
            T = T + shift - ((T + shift - window_length) % shift)
    return left, T


def num_frames(num_samples, window_length, shift, fading, pad):
    """Frames the conv1d of ``_stft.py:158`` produces ("valid" conv, stride=shift)."""
    _, T = padded_length(num_samples, window_length, shift, fading, pad)
    return (T - window_length) // shift + 1


# --------------------------------------------------------------------------- forward
def stft(x, size=1024, shift=256, *, window='blackman', window_length=None,
         fading='full', pad=True, symmetric_window=False):
    """STFT of ``x[..., T]`` -> complex ``[..., frames, size//2+1]`` (float64 maths).

    ``X[t, n] = sum_k x_p[t*shift + k] * w[k] * exp(-2j*pi*n*k/size)`` -- identical to the
    dense conv of ``_stft.py:11-23,158`` (and to ``paderbox.transform.stft``).
    """
    x = np.asarray(x, dtype=np.float64)
    L = size if window_length is None else window_length
    assert size % 2 == 0
    w = get_window(window, symmetric_window, L)
    left, T = padded_length(x.shape[-1], L, shift, fading, pad)
    xp = np.zeros(x.shape[:-1] + (max(T, left + x.shape[-1]),))
    xp[..., left:left + x.shape[-1]] = x
    n = (T - L) // shift + 1
    idx = np.arange(L)[None, :] + shift * np.arange(n)[:, None]
    frames = xp[..., idx] * w
    return np.fft.rfft(frames, n=size, axis=-1)


def stft_dense(x, size, shift, window, fading='full', pad=True):
    """Same transform through the explicit DFT*window matrix of ``_stft.py:11-23``.

    Independent of numpy's FFT; used to cross-check :func:`stft` on small sizes.
    ``window`` is the window *array*.
    """
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(window, dtype=np.float64)
    L = len(w)
    k = np.arange(L)
    n = np.arange(size // 2 + 1)[:, None]
    real = np.cos(-1 * n * 2 * np.pi / size * k) * w
    imag = np.sin(-1 * n * 2 * np.pi / size * k) * w
    left, T = padded_length(x.shape[-1], L, shift, fading, pad)
    xp = np.zeros(x.shape[:-1] + (max(T, left + x.shape[-1]),))
    xp[..., left:left + x.shape[-1]] = x
    nfr = (T - L) // shift + 1
    idx = k[None, :] + shift * np.arange(nfr)[:, None]
    fr = xp[..., idx]
    return fr @ real.T + 1j * (fr @ imag.T)


# --------------------------------------------------------------------------- inverse
def istft(X, size=1024, shift=256, *, window='blackman', window_length=None,
          fading='full', symmetric_window=False):
    """Inverse of :func:`stft` following ``_stft.py:176-263`` (no ``num_samples`` cut).

    frame_t = irfft(X_t, size)[:L] * ws ; overlap-add at hop ``shift``; cut the fading pad.
    (The reference's conv-transpose kernels implement ``size * irfft`` with ``ws / size``.)
    Imaginary parts of the DC and Nyquist bins are ignored, exactly like the
    ``sin(0) = sin(pi n) = 0`` columns of ``_stft.py:37-40``.
    """
    X = np.asarray(X)
    L = size if window_length is None else window_length
    ws = biorthogonal_window(get_window(window, symmetric_window, L), shift)
    fr = np.fft.irfft(X, n=size, axis=-1)[..., :L] * ws
    n = X.shape[-2]
    out = np.zeros(X.shape[:-2] + ((n - 1) * shift + L,))
    for t in range(n):
        out[..., t * shift:t * shift + L] += fr[..., t, :]
    if fading not in (None, False):
        pw = L - shift
        if fading == 'half':
            pw = pw / 2
        out = out[..., int(pw):out.shape[-1] - ceil(pw)]
    return out


def to_representation(X, complex_representation):
    """``_stft.py:162-174``: complex | concat ``[re..., im...]`` | stacked ``[..., 2]``."""
    if complex_representation == 'complex':
        return X
    if complex_representation == 'concat':
        return np.concatenate([X.real, X.imag], axis=-1)
    if complex_representation == 'stacked':
        return np.stack([X.real, X.imag], axis=-1)
    raise ValueError(complex_representation)
