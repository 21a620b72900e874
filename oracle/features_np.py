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
