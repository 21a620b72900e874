import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import padertorch_amd as pt
from bench_kernels import timeit
dev = torch.device('cuda:0')
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 64000)
g = torch.Generator().manual_seed(0)
s = (0.1 * torch.randn(B, 2, N, generator=g)).to(dev)
y = s.sum(1)
t = timeit(lambda: pt.ops.pit_features(y, s), iters=10)
frames = B * ((N + 384 + 127) // 128)
print(f"DBG={os.environ.get('PTMI_STFT_DBG')} NG={os.environ.get('PTMI_FEAT_NG')} features {t:.1f} us {frames * 6676 / t / 1e3:.0f} GB/s")
