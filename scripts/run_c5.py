import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
import bench_configs as bc
bc.tuning.use_tuned_gemms()
print(bc.dc(64, 16000, 4, 'C5'))
