"""paderbox.transform.module_stft stand-in: delegates to the repo's oracle (oracle/stft_np.py)."""
from oracle import stft_np as _o


def _get_window(window, symmetric_window, window_length):
    return _o.get_window(window, symmetric_window, window_length)


def _biorthogonal_window_fastest(analysis_window, shift, use_amplitude=False):
    assert not use_amplitude
    return _o.biorthogonal_window(analysis_window, shift)


def _samples_to_stft_frames(samples, size, shift, *, pad=True, fading=None):
    return _o.samples_to_frames(samples, size, shift, pad=pad, fading=fading)


def _stft_frames_to_samples(frames, size, shift, fading=None):
    return _o.frames_to_samples(frames, size, shift, fading=fading)


def sample_index_to_stft_frame_index(*args, **kwargs):
    raise NotImplementedError('parity unpinned in the reference tree')


def stft(time_signal, size=1024, shift=256, *, axis=-1, window='blackman', window_length=None,
         fading='full', pad=True, symmetric_window=False):
    assert axis in (-1,)
    return _o.stft(time_signal, size, shift, window=window, window_length=window_length,
                   fading=fading, pad=pad, symmetric_window=symmetric_window)


def istft(stft_signal, size=1024, shift=256, *, window='blackman', fading='full',
          window_length=None, symmetric_window=False, num_samples=None, pad=True,
          biorthogonal_window=None):
    out = _o.istft(stft_signal, size, shift, window=window, window_length=window_length,
                   fading=fading, symmetric_window=symmetric_window)
    if num_samples is not None:
        out = out[..., :num_samples]
    return out
