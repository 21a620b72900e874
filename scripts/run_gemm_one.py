#!/usr/bin/env python3
"""One split-GEMM shape a few times (profiling target): run_gemm_one.py M N K form [products]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from padertorch_amd.ops import gemm
M, N, K, form = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
prod = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
if form == 'nt':
    a, b = torch.randn(M, K, generator=g).to(dev), (0.05 * torch.randn(N, K, generator=g)).to(dev).t()
elif form == 'nn':
    a, b = torch.randn(M, K, generator=g).to(dev), (0.05 * torch.randn(K, N, generator=g)).to(dev)
else:
    a, b = torch.randn(K, M, generator=g).to(dev).t(), torch.randn(K, N, generator=g).to(dev)
out = torch.empty(M, N, device=dev)
ax, ay = gemm.absmax(a), gemm.absmax(b)
for _ in range(5):
    gemm.mm(a, b, out=out, amax_x=ax, amax_y=ay, products=prod, split_k=1)
torch.cuda.synchronize()
