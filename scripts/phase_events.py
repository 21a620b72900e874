#!/usr/bin/env python3
"""Where the critical path of the bench step goes WITHOUT a profiler attached (rocprofv3 slows the host down until the step head and
every cross-queue hand-over are host bound): HIP events around the few launches that mark the phases of a step - front-end, weight
preparation, recurrences, the GEMMs of the main queue, loss, norm, Adam - in the steps that do not carry bench.py's own kernel
events.  Times are relative to the start of the step's feature kernel, averaged over the steps.
    python scripts/phase_events.py [bench.py arguments]  ->  stdout (gpurun_out/phase_events.txt in scripts/refresh_profiles.sh)"""
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.argv = ['bench.py', '--steps', '48', '--warmup', '10', '--no-cpu-baseline', '--no-extras', '--eager'] + sys.argv[1:]       # (the eager step: a replay takes no event records)
import torch  # noqa: E402

from padertorch_amd import _lib  # noqa: E402

MARK = ('pit_features', 'lstm_forward', 'lstm_backward', 'adam_flat', 'grad_norm', 'pit_pairwise_sse', 'pit_backward', 'lstm_weight_prep',
        'gemm_planes', 'dc_loss_forward', 'dc_loss_backward')
records = []
_orig = _lib.timed


def timed(name, fn, *args):
    if _lib.KERNEL_TIMERS is not None or not name.startswith(MARK):
        return _orig(name, fn, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    records.append((name, torch.cuda.current_stream().cuda_stream, e0, e1))
    return rc


_lib.timed = timed
import bench  # noqa: E402

bench.main()
torch.cuda.synchronize()
# steps: from one pit_features to the next
steps, cur = [], None
for rec in records:
    if rec[0] == 'pit_features':
        cur = []
        steps.append(cur)
    if cur is not None:
        cur.append(rec)
steps = [s for s in steps if s[-1][0] == 'adam_flat'][5:]          # whole steps, warm
periods = [a[0][2].elapsed_time(b[0][2]) * 1e3 for a, b in zip(steps, steps[1:])]
shape = [tuple(r[0] for r in s) for s in steps]
common = max(set(shape), key=shape.count)
steps = [s for s, sh in zip(steps, shape) if sh == common]
main_stream = steps[0][0][1]
acc = defaultdict(lambda: [0., 0.])
for s in steps:
    base = s[0][2]
    for i, (name, stream, e0, e1) in enumerate(s):
        acc[i][0] += base.elapsed_time(e0) * 1e3
        acc[i][1] += e0.elapsed_time(e1) * 1e3
n = len(steps)
print(f'# feature kernel to feature kernel: median {sorted(periods)[len(periods) // 2]:.1f} us')
print(f'# {n} steps of {len(common)} marked launches; us from the start of the feature kernel; queue m = main, s = side')
prev_end = 0.
for i, name in enumerate(common):
    t, d = acc[i][0] / n, acc[i][1] / n
    q = 'm' if steps[0][i][1] == main_stream else 's'
    gap = f'{t - prev_end:8.1f}' if q == 'm' else '        '
    print(f'{q} {t:9.1f} {d:8.1f}  gap on main {gap}  {name}')
    if q == 'm':
        prev_end = t + d
