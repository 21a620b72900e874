"""lstm_weight_prep alone: us per call for the layer shapes of the c2 model (host bound at ~22 us: run it under `rocprofv3 --kernel-trace
--stats` for the kernel's own duration).  The PTMI_PREP_DBG ablation bits this script sweeps (1 no W_ih job, 2 no W_hh job, 4 no maximum word,
8 unconditional atomicMax) existed in csrc/lstm_prep.hip while profiles/r6_head_of_step.txt was measured and were removed afterwards."""
import os
import sys
import subprocess
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
if len(sys.argv) == 1:
    for dbg in (0, 1, 2, 4, 3, 7):
        subprocess.run([sys.executable, __file__, str(dbg)], env=dict(os.environ, PTMI_PREP_DBG=str(dbg)))
    sys.exit(0)
import torch  # noqa: E402
import padertorch_amd  # noqa: E402,F401
dev = torch.device('cuda', 0)
for I in (257, 1200):
    H = 600
    ps = [[torch.randn(4 * H, I, device=dev), torch.randn(4 * H, H, device=dev), torch.randn(4 * H, device=dev), torch.randn(4 * H, device=dev)] for _ in range(2)]
    args = ([p[0] for p in ps], [p[1] for p in ps], [p[2] for p in ps], [p[3] for p in ps], 608)
    for _ in range(5):
        torch.ops.ptmi.lstm_weight_prep(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        torch.ops.ptmi.lstm_weight_prep(*args)
    e1.record()
    torch.cuda.synchronize()
    print(f'dbg={sys.argv[1]} I={I}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per call (incl. allocation + zero_words)')
