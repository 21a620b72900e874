"""Oracle mel features (oracle/features_np.py) vs the reference goldens g8 (CPU)."""
import numpy as np

from oracle import features_np as O


def test_mel_transform_vs_reference(g8):
    for c in g8['configs']:
        key = c['key']
        fb = O.mel_fbanks_normalized(c['sample_rate'], c['stft_size'], c['number_of_filters'], c['lowest_frequency'],
                                     c['highest_frequency'], c['htk_mel'])
        np.testing.assert_allclose(fb, g8[f'{key}/fbanks'], rtol=0, atol=1e-7)     # reference normalisation
        y = O.mel_transform(g8[f'{key}/spec'], fb, log=c['log'])
        np.testing.assert_allclose(y, g8[f'{key}/mel'], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(O.mel_inverse(g8[f'{key}/mel'], fb, log=c['log']), g8[f'{key}/inverse'],
                                   rtol=2e-4, atol=1e-6)
        # filter shape properties that any mel filterbank has: non-negative, one contiguous band,
        # peaks ascending in frequency
        raw = O.get_fbanks(c['sample_rate'], c['stft_size'], c['number_of_filters'], c['lowest_frequency'],
                           c['highest_frequency'], c['htk_mel'])
        assert (raw >= 0).all() and raw.max() <= 1. + 1e-12
        peaks = raw.argmax(-1)
        assert (np.diff(peaks) >= 0).all()
        for row in raw:
            nz = np.flatnonzero(row)
            assert len(nz) == 0 or (np.diff(nz) == 1).all()


def test_front_end_vs_reference(g8):
    x = g8['front/x']
    fb = O.mel_fbanks_normalized(16000, 512, 80)
    np.testing.assert_allclose(O.logmel_from_waveform(x, fb, 512, 128), g8['front/logmel'], rtol=1e-4, atol=1e-4)
    fb1 = O.mel_fbanks_normalized(16000, 512, 40)
    np.testing.assert_allclose(O.logmel_from_waveform(x, fb1, 512, 128, log=False, power=1), g8['front/mel_magnitude'],
                               rtol=1e-4, atol=1e-6)
    fb2 = O.mel_fbanks_normalized(16000, 1024, 64)
    np.testing.assert_allclose(
        O.logmel_from_waveform(x, fb2, 1024, 256, window_length=800, window='hann', fading='half'),
        g8['front/logmel_1024_256_800'], rtol=1e-4, atol=1e-4)
