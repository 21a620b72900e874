import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench_kernels import timeit
from padertorch_amd.ops.losses import dc_loss_batched
dev = torch.device('cuda:0')
T, B, E, K, F = 503, 64, 20, 3, 257
x = torch.nn.functional.normalize(torch.randn(T, B, E, F, device=dev), dim=-2).requires_grad_(True)
tm = torch.nn.functional.one_hot(torch.randint(0, K, (B, T, F), device=dev), K).permute(0, 1, 3, 2).float().contiguous()
nbytes = T * B * F * (E + K) * 4
t = timeit(lambda: dc_loss_batched(x.detach(), tm), iters=10)
print(f"dc forward {t:.1f} us {nbytes / t / 1e3:.0f} GB/s")
loss = dc_loss_batched(x, tm)[0]
t = timeit(lambda: torch.autograd.grad(loss, x, retain_graph=True), iters=10)
print(f"dc backward {t:.1f} us {(nbytes + T * B * F * E * 4) / t / 1e3:.0f} GB/s")
