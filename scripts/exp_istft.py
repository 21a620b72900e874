import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import padertorch_amd as pt
from bench_kernels import timeit
dev = torch.device('cuda:0')
B, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 1536), (int(sys.argv[2]) if len(sys.argv) > 2 else 64000)
st = pt.ops.STFT(512, 128)
x = (0.1 * torch.randn(B, N)).to(dev)
X = st(x)
t = timeit(lambda: st.inverse(X), iters=10)
nbytes = X.numel() * 8 + B * N * 4
print(f"B={B} N={N} DBG={os.environ.get('PTMI_STFT_DBG')} istft {t:.1f} us {nbytes / t / 1e3:.0f} GB/s")
