import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.cuda.tunable as tunable
import sys; from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent))
import library_gemm_tuning as tuning
dev = torch.device('cuda:0')
x = torch.randn(8096, 1200, device=dev)
lin = torch.nn.Linear(1200, 4800).to(dev)


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('default   linear 8096x1200x4800: %.1f us' % t(lambda: lin(x)))
p = tuning.use_tuned_gemms()
print('file', p, 'enabled', tunable.is_enabled(), 'tuning', tunable.tuning_is_enabled(), 'filename', tunable.get_filename())
print('results loaded:', len(tunable.get_results()))
print('tuned     linear 8096x1200x4800: %.1f us' % t(lambda: lin(x)))
print('results after:', len(tunable.get_results()))
for r in tunable.get_results()[:3]:
    print(r)
