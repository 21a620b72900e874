import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import padertorch_amd as pt
from bench_kernels import timeit
dev = torch.device('cuda:0')
print('DBG', os.environ.get('PTMI_STFT_DBG'))
B, N = 1536, 64000
x = (0.1 * torch.randn(B, N)).to(dev)
st = pt.ops.STFT(512, 128)
t = timeit(lambda: st(x), iters=10)
print(f'stft_fwd {t:.1f} us')
if not os.environ.get('PTMI_STFT_DBG'):
    out = torch.empty(B, 503, 257, 2, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for fn, name, nbytes in [(lambda: out.fill_(1.0), 'fill 1.59GB', out.numel() * 4),
                             (lambda: out.copy_(out2), 'copy 1.59GB (r+w)', out.numel() * 8)]:
        out2 = torch.empty_like(out)
        fn(); torch.cuda.synchronize(); e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        print(f'{name}: {us:.1f} us  {nbytes / us / 1e3:.0f} GB/s')
