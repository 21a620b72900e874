import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from padertorch_amd import ops
dev = 'cuda:0'
N, E, F = 32192, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 257
x = torch.randn(N, E, F, device=dev, requires_grad=True)
g = torch.randn(N, E, F, device=dev)
def run(fn, n=10):
    for _ in range(3):
        y = fn(x); y.backward(g); x.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        y = fn(x)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(n):
        y = fn(x); y.backward(g); x.grad = None
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, ((t2 - t1) - (t1 - t0)) / n * 1e6
gb = N * E * F * 4 / 1e9
for name, fn in [('torch', lambda t: torch.nn.functional.normalize(t, dim=-2)), ('hip', ops.unit_norm)]:
    f, b = run(fn)
    print(f'{name}: fwd {f:.0f} us ({2 * gb / f * 1e6 / 1e3:.2f} TB/s of 2 passes), bwd {b:.0f} us ({3 * gb / b * 1e6 / 1e3:.2f} TB/s of 3 passes)')
