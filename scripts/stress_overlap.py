"""Repeated cold starts of several configurations with the side-stream weight gradients enabled (each
process includes an unsynchronised first step, the situation that hung with Stream-K side GEMMs)."""
import faulthandler
import torch
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
faulthandler.dump_traceback_later(150, exit=True)
import bench_configs as bc  # noqa: E402
from padertorch_amd.ops import lstm as _lstm  # noqa: E402

_lstm.DEFER_WGRAD = True
_lstm.warm_side_stream(bc.dev)
bc.tuning.use_tuned_gemms()
B, fs, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
r = (bc.pit if kind == 'pit' else bc.dc)(B, fs, 4, f'{kind}-B{B}-{fs}')
err = int(_lstm.error_word(torch.device('cuda:0')))
print(r['config'], round(r['ms_per_step'], 2), 'ms/step, spin errors', err, flush=True)
