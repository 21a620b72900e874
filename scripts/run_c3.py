import sys, faulthandler
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
faulthandler.dump_traceback_later(90, exit=True)
import bench_configs as bc
from padertorch_amd.ops import lstm as _lstm
_lstm.DEFER_WGRAD = '--no-overlap' not in sys.argv
bc.tuning.use_tuned_gemms()
print(bc.pit(64, 16000, 4, 'C3'), flush=True)
