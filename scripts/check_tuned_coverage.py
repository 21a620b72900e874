"""Are all GEMM shapes of the bench step covered by the committed TunableOp file?  Runs two steps with the
online search enabled and prints the entries the search had to add."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch.cuda.tunable as tunable
import bench_configs as bc
from padertorch_amd.ops import lstm as _lstm
_lstm.DEFER_WGRAD = True
_lstm.WGRAD_SIDE_STREAM = False
bc.tuning.use_tuned_gemms(search=True)
before = {(op, p) for op, p, s, t in tunable.get_results()}
bc.pit(32, 8000, 4, 'C2')
for op, p, s, t in tunable.get_results():
    if (op, p) not in before:
        print('NEW', op, p, s, round(t, 4))
print('entries before', len(before), 'after', len(tunable.get_results()))
