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


def example_to_device(example, device=None, memo=None):
    """Moves a nested structure to the device; numpy arrays become tensors.

    Objects already moved are tracked by ``id`` like ``copy.deepcopy`` so that an array referenced
    twice is transferred once (``data/batch.py:56-81``).  H2D copies are issued non-blocking.
    """
    from ..ops.sequence.pack_module import PaddedList
    if memo is None:
        memo = {}

    def convert(value):
        id_ = id(value)
        if id_ in memo:
            return memo[id_]
        if isinstance(value, np.ndarray):
            try:
                value = torch.from_numpy(value)
            except TypeError:
                if value.dtype not in [np.complex64, np.complex128]:
                    raise
        if isinstance(value, (torch.Tensor, PaddedList)):
            value = value.to(device) if isinstance(value, PaddedList) else \
                value.to(device=device, non_blocking=True)
        memo[id_] = value
        return value

    return _nested(convert, example)


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
