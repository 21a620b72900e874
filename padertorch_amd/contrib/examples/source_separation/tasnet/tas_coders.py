"""In-graph STFT encoder / iSTFT decoder of the reference's TasNet
(``padertorch/contrib/examples/source_separation/tasnet/tas_coders.py:138-240``) on the HIP STFT.

Same constructor arguments, shapes and values; both directions are differentiable (the adjoint
kernels of ``padertorch_amd.ops.STFT``).
"""
from typing import Tuple, Union

import torch
from einops import rearrange

from .....ops import STFT


class StftEncoder(torch.nn.Module):
    """``[..., T] -> [..., feature_size, frames]`` (real | imaginary parts stacked along features).

    ``StftEncoder(feature_size=258)(mixture[2, 6, 203], [203, 150])`` -> ``[2, 6, 258, 20]`` and
    ``num_frames == [20, 14]`` (reference doctest ``tas_coders.py:140-155``).
    """

    def __init__(self, window_length: int = 20, feature_size: int = 256, stride: int = None):
        super().__init__()
        self.window_length = window_length
        self.feature_size = feature_size
        self.stride = stride
        if stride is None:
            stride = window_length // 2
        # feature_size - 2 because the stft adds two uninformative values for an even size
        self.stft = STFT(size=feature_size - 2, shift=stride, window_length=window_length,
                         fading=False, complex_representation='concat')

    def forward(self, inputs, sequence_lengths: torch.Tensor = None
                ) -> Tuple[torch.Tensor, Union[torch.Tensor, None]]:
        encoded = self.stft(inputs)
        encoded = rearrange(encoded, '... frames fbins -> ... fbins frames')
        if sequence_lengths is not None:
            num_frames = torch.tensor([self.stft.samples_to_frames(int(samples)) for samples in sequence_lengths])
            return encoded, num_frames
        return encoded


class IstftDecoder(torch.nn.Module):
    """``[B, ..., feature_size, frames] -> [B, ..., T]`` (reference doctest ``tas_coders.py:197-209``:
    ``[2, 4, 258, 10] -> [2, 4, 110]``)."""

    def __init__(self, window_length: int = 20, feature_size: int = 256, stride: int = None):
        super().__init__()
        self.window_length = window_length
        self.feature_size = feature_size
        self.stride = stride
        if stride is None:
            stride = window_length // 2
        self.stft = STFT(size=feature_size - 2, window_length=window_length, shift=stride,
                         fading=False, complex_representation='concat')

    def forward(self, stft_signal) -> torch.Tensor:
        stft_signal = rearrange(stft_signal, '... fbins frames  -> ... frames fbins')
        return self.stft.inverse(stft_signal)
