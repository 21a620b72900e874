#!/usr/bin/env python3
"""BASELINE configs[1] "bf16" mode: the same PIT training step with the dense layers multiplying plain 16-bit operands (the hi
planes only, ops.gemm.PRODUCTS = 1: fp16 for activations and weights, bf16 for the gate gradients; fp32 accumulation) against
the default (three 16-bit products per product, fp32-equivalent), same weights, same inputs: delta of loss, masks and
gradients, and the time of both.  Reported, not asserted."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import padertorch_amd as pt  # noqa: E402
from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel  # noqa: E402
from padertorch_amd.ops import gemm  # noqa: E402

dev = torch.device('cuda:0')
B, n = 32, 32000
torch.manual_seed(0)
model = PermutationInvariantTrainingModel().to(dev).train()
g = torch.Generator().manual_seed(1)
s = (0.1 * torch.randn(B, 2, n, generator=g)).to(dev)
feats = pt.ops.pit_features(s.sum(1), s)


def step(products):
    gemm.PRODUCTS = products
    model.zero_grad(set_to_none=True)
    masks = model(feats)
    losses = model.review(feats, masks)['losses']
    losses['pit_ips_loss'].backward()
    torch.cuda.synchronize()
    return (torch.stack([m.detach() for m in masks]).clone(), {k: float(v) for k, v in losses.items()},
            {k: p.grad.detach().clone() for k, p in model.named_parameters()})


def timed(products, reps=10):
    gemm.PRODUCTS = products
    for _ in range(2):
        step(products)
    t0 = time.perf_counter()
    for _ in range(reps):
        step(products)
    return (time.perf_counter() - t0) / reps * 1e3


m3, l3, g3 = step(3)
m1, l1, g1 = step(1)
rel = {k: float((g1[k] - g3[k]).norm() / g3[k].norm().clamp_min(1e-30)) for k in g3}
out = dict(workload='PIT 3xBLSTM-600, B=32, T=253 (BASELINE configs[1] shape), forward + review + backward, no optimizer',
           reference='dense layers with 3 fp16 products per fp32 product (default, fp32-equivalent)',
           bf16='dense layers on the hi planes only (plain fp16 / bf16 operands, fp32 accumulation); recurrence unchanged',
           loss_default=l3, loss_bf16=l1, loss_delta={k: l1[k] - l3[k] for k in l3},
           mask_max_abs_delta=float((m1 - m3).abs().max()), mask_rms_delta=float((m1 - m3).pow(2).mean().sqrt()),
           grad_rel_l2_delta_max=max(rel.values()), grad_rel_l2_delta=rel,
           ms_fwd_bwd_default=timed(3), ms_fwd_bwd_bf16=timed(1))
gemm.PRODUCTS = 3
print(json.dumps(out))
