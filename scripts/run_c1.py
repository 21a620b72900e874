import sys, faulthandler
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
faulthandler.dump_traceback_later(60, exit=True)
import bench_configs as bc
from padertorch_amd.ops import lstm as _lstm
_lstm.DEFER_WGRAD = '--no-overlap' not in sys.argv
bc.tuning.use_tuned_gemms()
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
print(bc.pit(B, 8000, 4, f'B{B}'), flush=True)
