"""HIP (log-)mel kernels vs the reference goldens g8 and the oracle (GPU, through the C ABI)."""
import numpy as np
import pytest
import torch

from oracle import features_np as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_mel_transform_vs_reference(g8):
    from padertorch_amd.contrib.je.modules.features import MelTransform
    for c in g8['configs']:
        key = c['key']
        mt = MelTransform(c['sample_rate'], c['stft_size'], c['number_of_filters'],
                          lowest_frequency=c['lowest_frequency'], highest_frequency=c['highest_frequency'],
                          htk_mel=c['htk_mel'], log=c['log']).to(DEV)
        np.testing.assert_allclose(mt.fbanks.cpu().numpy(), g8[f'{key}/fbanks'], rtol=0, atol=1e-7)
        spec = torch.from_numpy(g8[f'{key}/spec']).to(DEV)
        y, maxima = mt(spec, return_maxima=True)
        assert list(y.shape) == list(g8[f'{key}/mel'].shape)
        np.testing.assert_allclose(y.cpu().numpy(), g8[f'{key}/mel'], rtol=1e-5, atol=1e-5)
        assert maxima.cpu().tolist() == g8[f'{key}/maxima'].tolist()
        np.testing.assert_allclose(mt.inverse(y).cpu().numpy(), g8[f'{key}/inverse'], rtol=2e-4, atol=1e-6)
        # a user-supplied filterbank (e.g. paderbox's own get_fbanks output) goes through unchanged
        raw = O.get_fbanks(c['sample_rate'], c['stft_size'], c['number_of_filters'], c['lowest_frequency'],
                           c['highest_frequency'], c['htk_mel'])
        mt2 = MelTransform(c['sample_rate'], c['stft_size'], c['number_of_filters'], log=c['log'], fbanks=raw)
        np.testing.assert_allclose(mt2(spec).cpu().numpy(), g8[f'{key}/mel'], rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError):
        mt(spec.cpu())                          # no CPU fallback
    with pytest.raises(NotImplementedError):
        MelTransform(16000, 512, 40, warping_fn=lambda *a, **k: None)


def test_fused_front_end_vs_reference(g8):
    from padertorch_amd.ops import STFT
    from padertorch_amd.contrib.je.modules.features import MelTransform, stft_logmel
    x = torch.from_numpy(g8['front/x']).to(DEV)
    got = stft_logmel(x, STFT(512, 128), MelTransform(16000, 512, 80))
    assert list(got.shape) == list(g8['front/logmel'].shape)
    np.testing.assert_allclose(got.cpu().numpy(), g8['front/logmel'], rtol=1e-4, atol=1e-4)
    got = stft_logmel(x, STFT(512, 128), MelTransform(16000, 512, 40, log=False), power=1)
    np.testing.assert_allclose(got.cpu().numpy(), g8['front/mel_magnitude'], rtol=1e-4, atol=1e-6)
    got = stft_logmel(x, STFT(1024, 256, window_length=800, window='hann', fading='half'), MelTransform(16000, 1024, 64))
    np.testing.assert_allclose(got.cpu().numpy(), g8['front/logmel_1024_256_800'], rtol=1e-4, atol=1e-4)
    # fused == unfused HIP path (|STFT|^2 -> MelTransform), leading dims, ragged rows
    st, mt = STFT(512, 128), MelTransform(16000, 512, 80).to(DEV)
    g = torch.Generator().manual_seed(1)
    xb = (0.1 * torch.randn(2, 3, 5000, generator=g)).to(DEV)
    fused = stft_logmel(xb, st, mt)
    unfused = mt(st(xb).abs() ** 2)
    assert fused.shape == unfused.shape == (2, 3, st.samples_to_frames(5000), 80)
    np.testing.assert_allclose(fused.cpu().numpy(), unfused.cpu().numpy(), rtol=1e-4, atol=1e-4)
    ns = [5000, 4100, 3333, 5000, 777, 2048]
    ragged = stft_logmel(xb, st, mt, num_samples=ns).reshape(6, -1, 80)
    for r, n in enumerate(ns):
        T_r = st.samples_to_frames(n)
        want = O.logmel_from_waveform(xb.reshape(6, -1)[r, :n].cpu().numpy(), mt.fbanks.cpu().numpy(), 512, 128)
        np.testing.assert_allclose(ragged[r, :T_r].cpu().numpy(), want, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(ragged[r, T_r:].cpu().numpy(), np.log(1e-12), rtol=1e-6)   # frames past the row's end


def test_full_size_properties():
    """BASELINE config 5 shape (64 x 4 s @ 16 kHz, 80 mel bins): scaling the waveform by a shifts the
    log-mel power features by 2 log a; mel energies are a convex combination of the spectrum."""
    from padertorch_amd.ops import STFT
    from padertorch_amd.contrib.je.modules.features import MelTransform, stft_logmel
    st, mt = STFT(512, 128), MelTransform(16000, 512, 80).to(DEV)
    g = torch.Generator().manual_seed(0)
    x = (0.1 * torch.randn(64, 64000, generator=g)).to(DEV)
    a = stft_logmel(x, st, mt)
    b = stft_logmel(3. * x, st, mt)
    assert a.shape == (64, 503, 80)
    np.testing.assert_allclose((b - a).cpu().numpy(), 2 * np.log(3.), atol=2e-4)
    lin = MelTransform(16000, 512, 80, log=False).to(DEV)
    e = stft_logmel(x, st, lin)
    P = st(x).abs() ** 2
    assert bool((e <= P.max(-1, keepdim=True).values * (1 + 1e-5)).all()) and bool((e >= 0).all())
