"""Padding masks (reference: ``padertorch/ops/sequence/mask.py:4-73``)."""
import torch

__all__ = ['compute_mask']


def compute_mask(x, sequence_lengths, batch_axis=0, sequence_axis=1):
    """1 for positions ``t < sequence_lengths[b]``, 0 for padding, expanded to ``x.shape``.

    ``sequence_lengths=None`` gives an all-ones mask.  (Index glue on the device; the HIP kernels
    form the same mask from ``sequence_lengths`` on the fly and never materialise it.)
    """
    if sequence_lengths is None:
        return torch.ones_like(x)
    batch_axis %= x.dim()
    sequence_axis %= x.dim()
    lengths = torch.as_tensor(sequence_lengths).long().to(x.device)
    shape_b = [1] * x.dim()
    shape_b[batch_axis] = -1
    shape_t = [1] * x.dim()
    shape_t[sequence_axis] = -1
    idx = torch.arange(x.shape[sequence_axis], device=x.device).reshape(shape_t)
    return (idx < lengths.reshape(shape_b)).float().expand(x.shape)
