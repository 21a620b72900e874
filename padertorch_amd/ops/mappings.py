"""String -> activation module dispatcher (``padertorch/ops/mappings.py:35-44``)."""
import torch

__all__ = ['ACTIVATION_FN_MAP']


class DispatchError(KeyError):
    pass


class _CallableDispatcher(dict):
    """A callable key is returned as is; otherwise a dict with a clearer error on a miss."""

    def __getitem__(self, item):
        if callable(item):
            return item
        try:
            return super().__getitem__(item)
        except KeyError:
            raise DispatchError(f'Invalid option {item!r}. Choose one of {sorted(self)}.') from None


ACTIVATION_FN_MAP = _CallableDispatcher(
    relu=torch.nn.ReLU,
    prelu=torch.nn.PReLU,
    leaky_relu=torch.nn.LeakyReLU,
    elu=torch.nn.ELU,
    tanh=torch.nn.Tanh,
    sigmoid=torch.nn.Sigmoid,
    softmax=torch.nn.Softmax,
    identity=torch.nn.Identity,
)
