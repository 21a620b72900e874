import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import padertorch_amd as pt
from bench_kernels import timeit
dev = torch.device('cuda:0')
B, N = 512, 64000
g = torch.Generator().manual_seed(0)
s = (0.1 * torch.randn(B, 2, N, generator=g)).to(dev)
y = s.sum(1)
T = 503
for name, fn in [('Y only', lambda: pt.ops.pit_features(y, None)),
                 ('K=1', lambda: pt.ops.pit_features(y, s[:, :1].contiguous())),
                 ('K=2', lambda: pt.ops.pit_features(y, s))]:
    t = timeit(fn, iters=20)
    print(name, f'{t:.1f} us')
