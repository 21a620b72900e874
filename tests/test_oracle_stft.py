"""Pin oracle/stft_np.py against the reference's known answers (G1) and the reference import (G2)."""
import numpy as np
import pytest

from oracle import stft_np


def test_cb_known_answer(g1):
    # padertorch/contrib/cb/transform.py:219-232
    c = g1['cb_stft']
    X = stft_np.stft(np.array(c['input'], dtype=np.float32), **c['kwargs'])
    np.testing.assert_allclose(X.real, c['real'], atol=1e-12)
    np.testing.assert_allclose(X.imag, c['imag'], atol=1e-12)


def test_periodic_window():
    np.testing.assert_allclose(stft_np.get_window('hann', False, 4), [0, .5, 1, .5], atol=1e-15)
    np.testing.assert_allclose(stft_np.get_window('hann', True, 5), [0, .5, 1, .5, 0], atol=1e-15)


def test_frame_counts(g1):
    # tests/test_ops/test_stft.py:44-70,139-165
    for fc in g1['frame_counts']:
        for n, fr in zip(fc['samples'], fc['frames']):
            assert stft_np.samples_to_frames(n, fc['window_length'], fc['shift'], pad=True,
                                             fading=fc['fading']) == fr
            assert stft_np.num_frames(n, fc['window_length'], fc['shift'], fc['fading'], True) == fr
            X = stft_np.stft(np.zeros(n), fc['size'], fc['shift'], window_length=fc['window_length'],
                             fading=fc['fading'])
            assert X.shape == (fr, fc['size'] // 2 + 1)


def test_doctest_shapes(g1):
    for d in g1['doctest_shapes']:
        if 'inp' in d:
            X = stft_np.stft(np.zeros(d['inp']), d['size'], d['shift'], window_length=d['window_length'])
            assert list(stft_np.to_representation(X, d['rep']).shape) == d['out']
        else:
            X = np.zeros(d['inverse_inp'][:-1] + [d['size'] // 2 + 1], dtype=complex)
            x = stft_np.istft(X, d['size'], d['shift'], window_length=d['window_length'])
            assert list(x.shape) == d['inverse_out']


def test_option_grid_vs_reference(g2):
    for c in g2['cases']:
        x = g2[c['x']]
        kw = dict(window=c['window'], window_length=c['window_length'], fading=c['fading'])
        X = stft_np.stft(x, c['size'], c['shift'], pad=c['pad'], **kw)
        ref = g2[c['name'] + '_X']
        assert X.shape == ref.shape, c
        np.testing.assert_allclose(X, ref, atol=1e-9, err_msg=str(c))
        assert X.shape[-2] == c['frames']
        assert stft_np.samples_to_frames(x.shape[-1], c['window_length'], c['shift'], pad=c['pad'],
                                         fading=c['fading']) == c['frames']
        xi = stft_np.istft(ref, c['size'], c['shift'], **kw)
        np.testing.assert_allclose(xi, g2[c['name'] + '_xi'], atol=1e-9, err_msg=str(c))
        assert stft_np.frames_to_samples(ref.shape[-2], c['window_length'], c['shift'],
                                         fading=c['fading']) == c['samples_back'] == xi.shape[-1]
        if c['fading'] == 'full' and c['pad']:
            # perfect reconstruction (tests/test_ops/test_stft.py:36-42), incl. L % shift != 0
            np.testing.assert_allclose(xi[..., :x.shape[-1]], x, atol=1e-9, err_msg=str(c))


def test_dense_dft_matches_fft(g2):
    c = g2['cases'][0]
    x = g2[c['x']][..., :700]
    w = stft_np.get_window('hann', False, 50)
    a = stft_np.stft(x, 64, 24, window='hann', window_length=50)
    b = stft_np.stft_dense(x, 64, 24, w)
    np.testing.assert_allclose(a, b, atol=1e-10)


def test_representations(g2):
    x = g2['x_s512_h128'].astype(np.float32)
    X = stft_np.stft(x, 512, 128)
    for rep in ['concat', 'stacked']:
        ref = g2[f'rep_{rep}_X']
        got = stft_np.to_representation(X, rep)
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, atol=2e-4)   # reference ran in fp32 here


@pytest.mark.parametrize('fading', ['full', 'half', None])
def test_fullsize_property_roundtrip(fading):
    # BASELINE full size (4 s @ 8 kHz): T = 253 frames, round trip exact with fading='full'
    rng = np.random.RandomState(0)
    x = 0.1 * rng.standard_normal(32000)
    X = stft_np.stft(x, 512, 128, fading=fading)
    if fading == 'full':
        assert X.shape == (253, 257)
        np.testing.assert_allclose(stft_np.istft(X, 512, 128)[:32000], x, atol=1e-12)
