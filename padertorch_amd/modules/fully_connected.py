"""``fully_connected_stack`` (reference: ``padertorch/modules/fully_connected.py:9-66``).

Dropout -> Linear -> activation per layer in one ``torch.nn.Sequential`` whose entries are named ``dropout_<i>``,
``linear_<i>``, ``<activation>_<i>`` - the names are the ``state_dict`` keys of the reference's checkpoints
(``fully_connected.linear_0.weight`` ...), so they are kept.  On MI355X tensors every ``Linear`` of the stack runs on the
planes GEMM of ``csrc/gemm_planes.hip`` (``ops.linear``: fp32-equivalent split products on the 16-bit matrix cores,
hand-written input / weight gradients); parameters stay ``torch.nn.Linear`` parameters.
"""
import torch
from torch import nn

from ..ops import gemm as _gemm
from ..ops import linear as _linear
from ..ops.mappings import ACTIVATION_FN_MAP

__all__ = ['fully_connected_stack', 'PlanesLinear']


class PlanesLinear(nn.Linear):
    """``torch.nn.Linear`` (same parameters, same ``state_dict``) whose product runs in ``ops.linear.linear`` when the
    input lives on the GPU; CPU tensors take torch's own path (the reference trainer's CPU ``test_run``)."""

    def forward(self, input):
        if input.is_cuda and input.dim() >= 2:
            x = input.reshape(-1, input.shape[-1])
            if _gemm.usable(x, self.weight):           # checked here: ops.linear.linear would call this module again otherwise
                return _linear.linear(self, x).view(*input.shape[:-1], self.out_features)
        return torch.nn.functional.linear(input, self.weight, self.bias)


def _widths(input_size, hidden_size, output_size):
    if hidden_size is None:
        inner = []
    elif isinstance(hidden_size, int):
        inner = [hidden_size]
    elif isinstance(hidden_size, (list, tuple)):
        inner = list(hidden_size)
    else:
        raise TypeError(hidden_size)
    return [input_size, *inner, output_size]


def fully_connected_stack(input_size: int, hidden_size, output_size: int, activation: str = 'relu', dropout: float = 0.5,
                          output_activation: str = None):
    """``hidden_size``: None (one layer), an int (two layers) or a list of widths; ``dropout`` is the forget probability in
    front of EVERY linear (also the first); ``activation`` behind every layer but the last, ``output_activation`` (None /
    ``'identity'``: nothing) behind the last."""
    assert input_size is not None, input_size
    assert output_size is not None, output_size
    widths = _widths(input_size, hidden_size, output_size)
    n_layers = len(widths) - 1
    stack = nn.Sequential()
    for i in range(n_layers):
        act = activation if i + 1 < n_layers else output_activation
        stack.add_module(f'dropout_{i}', nn.Dropout(dropout))
        stack.add_module(f'linear_{i}', PlanesLinear(widths[i], widths[i + 1]))
        if act is not None and act != 'identity':
            stack.add_module(f'{act}_{i}', ACTIVATION_FN_MAP[act]())
    return stack
