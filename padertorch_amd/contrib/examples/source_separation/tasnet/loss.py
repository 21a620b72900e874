"""The TasNet training loss (``padertorch/contrib/examples/source_separation/tasnet/model.py:154-176``)
from ONE pass over the separated signals."""
import torch

from .....ops.losses import regression


def tasnet_loss(inputs: dict, outputs: dict) -> dict:
    """``inputs['s']``: targets ``[B, K, T]`` (padded), ``inputs['num_samples']``: lengths,
    ``outputs['out']``: estimates ``[B, K, T]``.  Returns the batch means of the permutation
    invariant ``si-sdr`` / ``log-mse`` / ``log1p-mse`` losses like ``TasNet.loss``: the reference
    runs ``3 * B * K!`` loss evaluations on slices, here all of them come from one statistics pass
    (``ptmi_td_pair_stats``) and one gradient pass (``ptmi_td_lincomb``)."""
    s, x = inputs['s'], outputs['out']
    if not isinstance(s, torch.Tensor):
        s = torch.stack(list(s))
    per_example = regression.pit_td_losses(x, s.to(x.device), lengths=inputs['num_samples'])
    return {k: torch.mean(v[0]) for k, v in per_example.items()}
