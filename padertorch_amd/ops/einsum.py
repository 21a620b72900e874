"""``einsum`` with capital letters (``padertorch/ops/einsum.py:10-19``)."""
import string

import torch

__all__ = ['einsum']


def einsum(operation: str, *operands):
    """Allows capital letters and collects operands as in `np.einsum`."""
    free = sorted(set(string.ascii_lowercase) - set(operation))
    for capital in sorted(set(string.ascii_uppercase) & set(operation)):
        operation = operation.replace(capital, free.pop())
    return torch.einsum(operation, *operands)
