"""STFT front and back end of the reference's TasNet variant on the HIP STFT
(``padertorch/contrib/examples/source_separation/tasnet/tas_coders.py:138-240``).

``StftEncoder(window_length, feature_size, stride)``: ``[..., T] -> [..., feature_size, frames]`` with the real
parts of the ``feature_size / 2`` bins on top of the imaginary parts; ``IstftDecoder`` is its inverse.  Both are
differentiable (the adjoint kernels of ``padertorch_amd.ops.STFT``).  Known answers held by the reference's
doctests (``:140-155``, ``:197-209``) and checked in ``tests/test_gpu_td.py``:
``StftEncoder(feature_size=258)(x[2, 6, 203], [203, 150]) -> [2, 6, 258, 20]`` with ``num_frames == [20, 14]``;
``IstftDecoder(feature_size=258)(X[2, 4, 258, 10]) -> [2, 4, 110]``.
"""
import torch

from .....ops import STFT


class _StftCoder(torch.nn.Module):
    """Shared geometry: an STFT of size ``feature_size - 2`` (an even-sized transform has ``size / 2 + 1`` bins, i.e.
    ``size + 2`` real values per frame), hop ``stride`` (default: half a window), no fading, (re | im) concatenated."""

    def __init__(self, window_length: int = 20, feature_size: int = 256, stride: int = None):
        super().__init__()
        self.window_length, self.feature_size, self.stride = window_length, feature_size, stride
        self.stft = STFT(size=feature_size - 2, shift=window_length // 2 if stride is None else stride,
                         window_length=window_length, fading=False, complex_representation='concat')


class StftEncoder(_StftCoder):
    def forward(self, inputs, sequence_lengths: torch.Tensor = None):
        """Returns the encoded signal, and the frame count of every ``sequence_lengths`` entry when those are given."""
        encoded = self.stft(inputs).transpose(-1, -2)               # frames x bins -> bins x frames
        if sequence_lengths is None:
            return encoded
        return encoded, torch.tensor([self.stft.samples_to_frames(int(n)) for n in sequence_lengths])


class IstftDecoder(_StftCoder):
    def forward(self, stft_signal) -> torch.Tensor:
        return self.stft.inverse(stft_signal.transpose(-1, -2))
