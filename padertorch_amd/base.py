"""``Module`` / ``Model``: the forward + review API that ``padertorch.Trainer`` drives.

Mirrors ``padertorch/base.py:55-73`` (``Module``) and ``:228-380`` (``Model``): abstract ``forward``
and ``review(inputs, outputs) -> dict`` with ``loss`` xor ``losses`` (+ optional ``scalars``,
``histograms``, ``audios``, ``images``, ``texts``, ``figures``, ``buffers``, ``snapshots``),
``modify_summary``, ``example_to_device`` and the ``create_snapshot`` flag.  Models written against
this base are plain ``torch.nn.Module`` s, so they run under the reference ``pt.Trainer`` unchanged
where padertorch is installed, and under :class:`padertorch_amd.train.Trainer` here.

Out of scope (reused from padertorch where needed): the ``Configurable``
factory/config system and the ``from_storage_dir`` checkpoint loaders.  Constructors take plain
JSON-able kwargs so the classes stay ``Configurable``-compatible.
"""
import abc

import numpy as np
import torch
from torch import nn

from .data.batch import example_to_device

__all__ = ['Module', 'Model']


class Module(nn.Module, abc.ABC):
    """Abstract base class for Modules (``base.py:55-73``)."""
    training: bool

    @abc.abstractmethod
    def forward(self, *args, **kwargs):  # pylint: disable=arguments-differ
        """Define the I/O behavior of Module()."""

    def load_checkpoint(self, checkpoint_path, in_checkpoint_path='model', map_location='cpu',
                        strict=True):
        """Load weights from a trainer checkpoint (subset of ``base.py:75-125``)."""
        ckpt = torch.load(str(checkpoint_path), map_location=map_location, weights_only=False)
        for part in (in_checkpoint_path.split('.') if in_checkpoint_path else []):
            ckpt = ckpt[part]
        self.load_state_dict(ckpt, strict=strict)
        return self


class Model(Module, abc.ABC):
    """Abstract base class for trainable models (``base.py:228-380``)."""

    # True when the model should create a snapshot (images, audios, ...) in the review
    create_snapshot: bool = False

    @abc.abstractmethod
    def forward(self, inputs):  # pylint: disable=arguments-differ
        """Single example (= one collated batch) -> whatever ``review`` expects."""

    @abc.abstractmethod
    def review(self, inputs, outputs):
        """Review dict with ``loss`` or ``losses`` and optional summary sub-dicts."""

    def modify_summary(self, summary):
        """``base.py:320-358``: scalars are averaged; buffers/snapshots must be consumed."""
        for key, scalar in summary['scalars'].items():
            summary['scalars'][key] = np.mean(scalar)
        assert len(summary.get('buffers', {})) == 0, \
            'intermediate format buffers has to be converted during modify_summary'
        assert len(summary.get('snapshots', {})) == 0, \
            'intermediate format snapshots has to be converted during modify summary'
        return summary

    def example_to_device(self, example, device=None, memo=None):
        """``base.py:360-380``."""
        return example_to_device(example, device, memo)
