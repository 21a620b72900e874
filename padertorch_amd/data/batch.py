"""``example_to_device`` / ``Sorter`` (``padertorch/data/batch.py:16-81,133-158``)."""
import dataclasses
import operator
from dataclasses import dataclass
from typing import Iterable, Union

import numpy as np
import torch

__all__ = ['example_to_device', 'Sorter']


def _nested(func, value):
    """Recursion over dict / list / tuple / dataclass (paderbox ``nested_op`` semantics)."""
    from ..ops.sequence.pack_module import PaddedList
    if isinstance(value, PaddedList):
        return func(value)
    if isinstance(value, dict):
        return value.__class__({k: _nested(func, v) for k, v in value.items()})
    if isinstance(value, (list, tuple)):
        return value.__class__([_nested(func, v) for v in value])
    if dataclasses.is_dataclass(value) and not isinstance(value, type):
        return value.__class__(**{f.name: _nested(func, getattr(value, f.name))
                                  for f in dataclasses.fields(value)})
    return func(value)


def _to_device(value, device):
    """One leaf of an example on ``device``.  numpy arrays become tensors first (complex ones included; a dtype torch has no
    counterpart for - strings, objects - raises ``torch.from_numpy``'s ``TypeError`` as in the reference), tensors cross with
    a non-blocking copy, a ``PaddedList`` moves its ONE padded buffer; everything else is not device data and stays."""
    from ..ops.sequence.pack_module import PaddedList
    if isinstance(value, PaddedList):
        return value.to(device)
    if isinstance(value, np.ndarray):
        value = torch.from_numpy(value)
    if isinstance(value, torch.Tensor):
        return value.to(device=device, non_blocking=True)
    return value


def example_to_device(example, device=None, memo=None):
    """Moves a nested structure to the device; numpy arrays become tensors (``data/batch.py:16-81``).

    ``memo`` maps ``id(host object) -> moved object`` across calls like ``copy.deepcopy``'s: an array that is referenced
    twice (inside one example, or by two examples that share the memo) crosses PCIe once and both places get the SAME
    tensor.  H2D copies are issued non-blocking.
    """
    seen = {} if memo is None else memo

    def moved(leaf):
        key = id(leaf)
        if key not in seen:
            seen[key] = _to_device(leaf, device)
        return seen[key]

    return _nested(moved, example)


@dataclass
class Sorter:
    """Sorts the examples of a batch by ``key`` (descending: required by ``pack_sequence``)."""
    key: Union[str, callable] = 'num_samples'
    reverse: bool = True

    def __post_init__(self):
        if not callable(self.key):
            self.key = operator.itemgetter(self.key)

    def __call__(self, examples: Iterable) -> tuple:
        return tuple(sorted(examples, key=self.key, reverse=self.reverse))
