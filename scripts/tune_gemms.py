#!/usr/bin/env python3
"""One-off TunableOp search over the GEMM shapes of the BASELINE configurations (run on the GPU box):

    python scripts/tune_gemms.py gpurun_out/tune/gfx950.csv

Starts from the committed selections, runs a few optimizer steps of every configuration with the
search enabled and writes the merged result file (commit it as scripts/tuned/<name>.csv).
"""
import os
import sys
from pathlib import Path

if '--overlap' in sys.argv:
    # shapes of the side-stream weight-gradient path: rocBLAS candidates only (no Stream-K kernels: they
    # must not spin on sibling workgroups next to a persistent recurrence kernel)
    os.environ['PYTORCH_TUNABLEOP_HIPBLASLT_ENABLED'] = '0'

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch.cuda.tunable as tunable  # noqa: E402

import bench_configs as bc  # noqa: E402

out = Path(sys.argv[1])
out.parent.mkdir(parents=True, exist_ok=True)
from padertorch_amd.ops import lstm as _lstm  # noqa: E402
_lstm.DEFER_WGRAD = '--overlap' in sys.argv     # also tune the per-direction shapes of the experimental side-stream path
_lstm.WGRAD_SIDE_STREAM = False                 # ... timed on the main stream, nothing else running
start = bc.tuning.DEFAULT_FILE
if '--overlap' in sys.argv:     # rocBLAS-only search from scratch (hipBLASLt solution ids cannot be loaded with
    import tempfile             # that backend switched off); merge its n = 4H "nt" entries into the committed file
    start = Path(tempfile.mkdtemp()) / 'empty.csv'
bc.tuning.use_tuned_gemms(results_file=start, search=True)
for fn, args in ((bc.pit, (4, 8000, 4, 'C1')), (bc.pit, (32, 8000, 4, 'C2')), (bc.pit, (64, 16000, 4, 'C3')),
                 (bc.dc, (64, 16000, 4, 'C5')), (bc.dc, (32, 8000, 4, 'DC-B32')))[int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 0:]:
    print(fn(*args), flush=True)
with open(out, 'w') as f:          # same layout TunableOp writes on exit
    for k, v in tunable.get_validators():
        f.write(f'Validator,{k},{v}\n')
    for op, params, solution, t in tunable.get_results():
        f.write(f'{op},{params},{solution},{t}\n')
print('wrote', out, len(tunable.get_results()), 'entries')
