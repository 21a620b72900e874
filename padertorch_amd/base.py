"""``Module`` / ``Model``: the forward + review API that ``padertorch.Trainer`` drives.

Mirrors ``padertorch/base.py:55-73`` (``Module``) and ``:228-380`` (``Model``): abstract ``forward``
and ``review(inputs, outputs) -> dict`` with ``loss`` xor ``losses`` (+ optional ``scalars``,
``histograms``, ``audios``, ``images``, ``texts``, ``figures``, ``buffers``, ``snapshots``),
``modify_summary``, ``example_to_device`` and the ``create_snapshot`` flag.  Models written against
this base are plain ``torch.nn.Module`` s, so they run under the reference ``pt.Trainer`` unchanged
where padertorch is installed, and under :class:`padertorch_amd.train.Trainer` here.

Out of scope (reused from padertorch where needed): the ``Configurable``
factory/config system.  Constructors take plain JSON-able kwargs so the classes stay ``Configurable``-compatible;
``Module.from_storage_dir`` / ``from_config_and_checkpoint`` (``base.py:83-225``) read the config
files the reference writes - nested ``{'factory': 'dotted.path', **kwargs}`` / ``{'partial': ...}`` dicts - and resolve
``padertorch.*`` factories to this package's classes, so a storage directory of the reference trainer loads here.
"""
import abc
import functools
import importlib
import json
from pathlib import Path

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


def _resolve(path):
    """Dotted path -> object.  ``padertorch.x.y.Z`` is looked up as ``padertorch_amd.x.y.Z`` first (the classes of the hot
    path exist under the same sub-paths here), then as written."""
    if not isinstance(path, str):
        return path
    candidates = [path]
    if path == 'padertorch' or path.startswith('padertorch.'):
        candidates.insert(0, 'padertorch_amd' + path[len('padertorch'):])
    last = None
    for name in candidates:
        parts = name.split('.')
        for cut in range(len(parts) - 1, 0, -1):
            try:
                obj = importlib.import_module('.'.join(parts[:cut]))
            except ImportError as e:
                last = e
                continue
            try:
                for attr in parts[cut:]:
                    obj = getattr(obj, attr)
                return obj
            except AttributeError as e:
                last = e
                break
    raise ImportError(f'cannot resolve the factory {path!r}: {last}')


def _instantiate(config):
    """A config node of the reference's ``Configurable`` files -> object: ``{'factory': f, **kw}`` calls ``f(**kw)``,
    ``{'partial': f, **kw}`` gives ``functools.partial(f, **kw)``, containers are walked, everything else is a value."""
    if isinstance(config, dict):
        if 'factory' in config or 'partial' in config:
            key = 'factory' if 'factory' in config else 'partial'
            assert not ('factory' in config and 'partial' in config), config
            fn = _resolve(config[key])
            kwargs = {k: _instantiate(v) for k, v in config.items() if k != key}
            return fn(**kwargs) if key == 'factory' else functools.partial(fn, **kwargs)
        return {k: _instantiate(v) for k, v in config.items()}
    if isinstance(config, (list, tuple)):
        return type(config)(_instantiate(v) for v in config)
    return config


def _from_config_and_checkpoint(cls, config_path, checkpoint_path, in_config_path='trainer.model', in_checkpoint_path='model',
                                map_location='cpu', strict=True):
    """``base.py:83-181``: build the module the config describes under ``in_config_path`` and load its weights from the
    checkpoint's ``in_checkpoint_path`` (MPI broadcast of the reference: not applicable, one process per GPU reads)."""
    config_path = Path(config_path)
    text = config_path.read_text()
    if config_path.suffix in ('.yaml', '.yml'):
        import yaml
        config = yaml.safe_load(text)
    else:
        config = json.loads(text)
    for part in (in_config_path.split('.') if in_config_path else []):
        config = config[part]
    module = _instantiate(config)
    assert isinstance(module, torch.nn.Module), type(module)
    return Module.load_checkpoint(module, checkpoint_path, in_checkpoint_path, map_location, strict)


def _from_storage_dir(cls, storage_dir, config_name='config.json', checkpoint_name='ckpt_best_loss.pth',
                      in_config_path='trainer.model', in_checkpoint_path='model', map_location='cpu', strict=True):
    """``base.py:183-225``: ``storage_dir / config_name`` + ``storage_dir / 'checkpoints' / checkpoint_name``."""
    storage_dir = Path(storage_dir)
    return cls.from_config_and_checkpoint(storage_dir / config_name, storage_dir / 'checkpoints' / checkpoint_name,
                                          in_config_path, in_checkpoint_path, map_location, strict)


Module.from_config_and_checkpoint = classmethod(_from_config_and_checkpoint)
Module.from_storage_dir = classmethod(_from_storage_dir)


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
