"""padertorch_amd: the padertorch PIT/deep-clustering hot path, MI355X-native (gfx950).

``import padertorch_amd as pt`` exposes the slice of the padertorch namespace the hot path uses:
``pt.ops.STFT``, ``pt.ops.losses.pit_loss`` / ``deep_clustering_loss``, ``pt.ops.pack_sequence`` ...,
``pt.Module`` / ``pt.Model``, ``pt.Trainer``, ``pt.optimizer.Adam``, ``pt.data.example_to_device``.
The arithmetic runs in hand-written HIP kernels (``csrc/``) behind the C ABI of ``include/ptmi.h``;
there is no CPU fallback.
"""
from . import data  # noqa: F401
from . import ops  # noqa: F401
from . import base  # noqa: F401
from .base import Module, Model  # noqa: F401
from . import modules  # noqa: F401
from . import train  # noqa: F401
from .train import optimizer  # noqa: F401
from .train.trainer import Trainer  # noqa: F401
from .ops import mappings  # noqa: F401
