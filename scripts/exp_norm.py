import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench_kernels import timeit
from padertorch_amd.modules import normalize
x = torch.randn(64, 503, 257, device="cuda:0", requires_grad=True)
lens = [503 - 3 * b for b in range(64)]
gamma = torch.ones(1, 1, 257, device="cuda:0", requires_grad=True); beta = torch.zeros(1, 1, 257, device="cuda:0", requires_grad=True)
nb = x.numel() * 4
t = timeit(lambda: normalize(x.detach(), gamma.detach(), beta.detach(), [1], 0, 1, lens, True, True, 1e-5))
print(f"normalize fwd btf/t (64x503x257): {t:.1f} us, {3 * nb / t / 1e3:.0f} GB/s (2 reads + 1 write)")
y = normalize(x, gamma, beta, [1], 0, 1, lens, True, True, 1e-5)[0]
t = timeit(lambda: torch.autograd.grad(y.sum(), (x, gamma, beta), retain_graph=True))
print(f"normalize bwd: {t:.1f} us, {7 * nb / t / 1e3:.0f} GB/s (6 reads + 1 write)")
x4 = torch.randn(32, 8, 64, 500, device="cuda:0")
t = timeit(lambda: normalize(x4, None, None, [0, 2, 3], 0, 3, None, True, True, 1e-5))
print(f"normalize fwd bcft/bft (32x8x64x500): {t:.1f} us, {3 * x4.numel() * 4 / t / 1e3:.0f} GB/s")
from padertorch_amd import _lib
import collections
for name, fn in (('fwd', lambda: normalize(x.detach(), gamma.detach(), beta.detach(), [1], 0, 1, lens, True, True, 1e-5)),
                 ('bwd', lambda: torch.autograd.grad(y.sum(), (x, gamma, beta), retain_graph=True))):
    fn(); torch.cuda.synchronize()
    _lib.KERNEL_TIMERS = []
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    acc = collections.defaultdict(list)
    for n, a, b in _lib.KERNEL_TIMERS:
        acc[n].append(a.elapsed_time(b) * 1e3)
    _lib.KERNEL_TIMERS = None
    print(name, {k: [round(v, 1) for v in vs[-4:]] for k, vs in acc.items()})
