"""numpy restatement of the PIT example's feature pipeline.  ORACLE - test infrastructure only.

Follows ``padertorch/contrib/examples/source_separation/pit/data.py:49-77``
(``pre_batch_transform``): ``S = stft(s, 512, 128)``, ``Y = stft(y, 512, 128)`` with the
paderbox defaults (blackman, fading='full', pad=True), then
``X_abs = |S|`` (T,K,F), ``Y_abs = |Y|`` (T,F),
``cos_phase_difference = cos(angle(Y)[:, None, :] - angle(X))`` (T,K,F), cast to float32.
Also ``Sorter`` (``padertorch/data/batch.py:133-158``) and ``collate_fn``
(``padertorch/data/utils.py:21-69``) for the dict-of-lists batch contract.
"""
import numpy as np

from . import stft_np


def pre_batch_transform(s, y, size=512, shift=128):
    """``s``: (K, N) sources, ``y``: (N,) mixture -> dict of float32 features (data.py:52-75)."""
    S = stft_np.stft(s, size, shift)            # (K, T, F)
    Y = stft_np.stft(y, size, shift)            # (T, F)
    X = np.transpose(S, (1, 0, 2))              # 'k t f -> t k f'
    return dict(
        s=np.ascontiguousarray(s, np.float32),
        y=np.ascontiguousarray(y, np.float32),
        Y=np.ascontiguousarray(Y, np.complex64),
        X_abs=np.ascontiguousarray(np.abs(X), np.float32),
        Y_abs=np.ascontiguousarray(np.abs(Y), np.float32),
        num_frames=Y.shape[0],
        cos_phase_difference=np.ascontiguousarray(
            np.cos(np.angle(Y[:, None, :]) - np.angle(X)), np.float32),
    )


def sort_and_collate(examples, key='num_frames'):
    """Sorter(key) (descending) followed by collate_fn: list of dicts -> dict of lists."""
    examples = sorted(examples, key=lambda e: e[key], reverse=True)
    return {k: [e[k] for e in examples] for k in examples[0]}


def synthetic_mixture(rng, num_samples, K=2, scale=0.1):
    """SURVEY.md section 8d synthetic input: K sources 0.1*N(0,1), mixture = sum."""
    s = (scale * rng.standard_normal((K, num_samples))).astype(np.float32)
    y = s.sum(axis=0).astype(np.float32)
    return s, y


# ---- (log-)mel features: padertorch/contrib/je/modules/features.py:214-339 (MelTransform) and the
# extractor front-end ``:171-176`` (power of the stacked STFT -> MelTransform) ------------------------
# The filterbank itself comes from the third-party ``paderbox.transform.module_fbank.get_fbanks``,
# which is absent here and exercised by no reference test in this tree:
# FILTERBANK PARITY TO PADERBOX IS UNPINNED.  It is restated from paderbox's published behaviour:
# HTK mel scale (or Slaney's with htk_mel=False), number_of_filters + 2 boundaries equally spaced in
# mel between lowest and highest frequency, triangles evaluated at the exact FFT bin frequencies.
# Everything downstream of the matrix (row normalisation, matmul, log, inverse) is the reference's
# own code and is pinned by the goldens.
def hz2mel(f, htk_mel=True):
    f = np.asarray(f, np.float64)
    if htk_mel:
        return 2595. * np.log10(1. + f / 700.)
    f_sp = 200. / 3.
    min_log_hz, logstep = 1000., np.log(6.4) / 27.
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)


def mel2hz(m, htk_mel=True):
    m = np.asarray(m, np.float64)
    if htk_mel:
        return 700. * (10. ** (m / 2595.) - 1.)
    f_sp = 200. / 3.
    min_log_hz, logstep = 1000., np.log(6.4) / 27.
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def get_fbanks(sample_rate, stft_size, number_of_filters, lowest_frequency=0., highest_frequency=None,
               htk_mel=True, warping_fn=None, size=()):
    """``[number_of_filters, stft_size // 2 + 1]`` float64 triangular mel filters with unit peak."""
    assert warping_fn is None and tuple(size) == (), 'frequency warping is a CPU augmentation: out of scope'
    if highest_frequency is None:
        highest_frequency = sample_rate / 2
    lowest_frequency = 0. if lowest_frequency is None else lowest_frequency
    if lowest_frequency < 0:
        lowest_frequency = sample_rate / 2 + lowest_frequency
    if highest_frequency < 0:
        highest_frequency = sample_rate / 2 + highest_frequency
    bounds = mel2hz(np.linspace(hz2mel(lowest_frequency, htk_mel), hz2mel(highest_frequency, htk_mel),
                                number_of_filters + 2), htk_mel)
    freqs = np.arange(stft_size // 2 + 1, dtype=np.float64) / stft_size * sample_rate
    lower, center, upper = bounds[:-2, None], bounds[1:-1, None], bounds[2:, None]
    up = (freqs[None] - lower) / (center - lower)
    down = (upper - freqs[None]) / (upper - center)
    return np.maximum(0., np.minimum(up, down))


def mel_fbanks_normalized(sample_rate, stft_size, number_of_filters, lowest_frequency=50., highest_frequency=None,
                          htk_mel=True):
    """The ``[F, M]`` float32 matrix MelTransform multiplies with (features.py:284-295)."""
    fbanks = get_fbanks(sample_rate, stft_size, number_of_filters, lowest_frequency, highest_frequency,
                        htk_mel).astype(np.float32)
    fbanks = fbanks / (fbanks.sum(axis=-1, keepdims=True) + 1e-6)
    return np.ascontiguousarray(fbanks.T)


def mel_transform(x, fbanks, log=True, eps=1e-12):
    """features.py:297-330: ``x [..., F] @ fbanks [F, M]``, then ``log(x + eps)``."""
    y = np.asarray(x, np.float64) @ np.asarray(fbanks, np.float64)
    return np.log(y + eps) if log else y


def mel_inverse(x, fbanks, log=True):
    """features.py:332-339."""
    ifbanks = np.asarray(fbanks, np.float64).T
    ifbanks = ifbanks / (ifbanks.sum(axis=-2, keepdims=True) + 1e-6)
    x = np.exp(x) if log else np.asarray(x, np.float64)
    return np.maximum(x @ ifbanks, 0.)


def logmel_from_waveform(x, fbanks, size=512, shift=128, log=True, eps=1e-12, power=2, **stft_kwargs):
    """Waveform -> STFT -> |X|^power -> mel -> log: the extractor front-end (features.py:171-176 with
    the stacked STFT of ``pt.ops.STFT``)."""
    X = stft_np.stft(x, size, shift, **stft_kwargs)
    return mel_transform(np.abs(X) ** power, fbanks, log=log, eps=eps)
