"""The N > 1 code path of ``bench.py`` end to end on a ONE-GPU box: two ranks under ``torch.distributed.run`` share cuda:0
(``PTMI_BENCH_SHARE_GPU=1``: collectives through gloo) with a small BLSTM (``PTMI_BENCH_UNITS=48``: two ranks' persistent recurrences
fit the chip side by side).  Functional only - what the driver's SCALE run exercises on real hardware: the probe of the three schedules
(the captured data-parallel step ``graph_split``, bucketed overlap, one all-reduce), the timed region on the chosen one, ``rccl.per_rank``."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def test_bench_two_ranks_probe_all_schedules_and_time_the_captured_step():
    env = dict(os.environ, PTMI_BENCH_SHARE_GPU='1', PTMI_BENCH_UNITS='48', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(REPO))
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(line) == 1, p.stdout[-2000:]
    d = json.loads(line[0])
    assert d['n_gpus'] == 2 and d['value'] > 0 and 'PRE-FLIGHT' in d['data']
    sched = d['rccl']['schedule']
    assert set(sched['probe_ms_per_step']) == {'graph_split', 'overlap', 'no_overlap'}, sched
    assert all(v is not None and v > 0 for v in sched['probe_ms_per_step'].values()), sched
    assert sched['used'] == min(sched['probe_ms_per_step'], key=sched['probe_ms_per_step'].get)
    ranks = d['rccl']['per_rank']
    assert [r['rank'] for r in ranks] == [0, 1]
    if sched['used'] == 'graph_split':
        for r in ranks:
            parts = r['gpu_parts']
            assert parts['steps'] == 4 and parts['graph_a_ms'] > 0 and parts['exchange_ms'] > 0 and parts['graph_b_ms'] > 0, r
        assert 'two hipGraphs' in d['config']['step_driver']
    assert d['roofline'] is not None and d['roofline']['frac'] > 0
