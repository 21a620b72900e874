#!/usr/bin/env python3
"""Launch ONE kernel family a few times (for rocprofv3 --pmc / --kernel-trace runs).

    python scripts/run_one.py {stft|istft|features|pit} [B] [N] [iters]
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import padertorch_amd as pt  # noqa: E402
from padertorch_amd.ops.losses import pit_mse_ips_losses  # noqa: E402

what = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = int(sys.argv[3]) if len(sys.argv) > 3 else 64000
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
s = (0.1 * torch.randn(B, 2, N, generator=g)).to(dev)
y = s.sum(1)
st = pt.ops.STFT(512, 128)
if what == 'stft':
    x3 = torch.cat([y[:, None], s], 1).reshape(-1, N).contiguous()
    for _ in range(iters):
        st(x3)
elif what == 'istft':
    X = st(torch.cat([y[:, None], s], 1).reshape(-1, N).contiguous())
    for _ in range(iters):
        st.inverse(X)
elif what == 'features':
    for _ in range(iters):
        pt.ops.pit_features(y, s)
elif what == 'pit':
    f = pt.ops.pit_features(y, s)
    mask = torch.rand(B, f['Y_abs'].padded.shape[1], 2, 257, device=dev, requires_grad=True)
    for _ in range(iters):
        loss = pit_mse_ips_losses(mask, f['Y_abs'].padded, f['X_abs'].padded,
                                  f['cos_phase_difference'].padded)[0]
        loss[1].backward()
torch.cuda.synchronize()
