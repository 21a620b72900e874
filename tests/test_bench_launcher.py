"""``python bench.py --gpus 2`` launches itself as two ranks (torch.distributed.run, 127.0.0.1) - here in --dry mode on
CPU with gloo: the same launcher, process group, layer buckets of the flat gradient buffer, all-reduce schedule and
JSON line as on the GPUs, around a stub step whose summed gradients are checked on every rank."""
import json
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def _run(*extra, **env_extra):
    env = dict(os.environ, OMP_NUM_THREADS='4', **env_extra)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--dry', '--steps', '2', '--warmup', '1', *extra],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(REPO))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_self_launch_two_ranks_bucketed_allreduce():
    out = _run('--gpus', '2')
    assert out['n_gpus'] == 2 and out['dry'] is True and out['steps'] == 2
    assert out['config']['parallelism'] == 'dp2' and out['config']['global_batch'] == 64
    r = out['rccl']
    assert r['world_size'] == 2 and r['overlap_allreduce'] is True
    assert r['schedule']['used'] == 'overlap' and set(r['schedule']['probe_ms_per_step']) == {'overlap', 'no_overlap'}
    # PIT model: three BLSTM layers and two linears -> five layer buckets covering all 23 480 914 parameters
    assert len(r['buckets']) == 5 and sum(r['buckets']) == 23480914 and r['flat_gradient_bytes'] == 4 * 23480914
    assert out['value'] > 0 and out['unit'] == 'frames/s'
    # what the first real SCALE record is read against: every bucket's own all-reduce, and the efficiency fields (the dry run has no
    # step without exchange to relate them to)
    assert len(r['bucket_all_reduce_ms']) == 5 and all(t > 0 for t in r['bucket_all_reduce_ms'])
    assert set(r['scaling']) == {'ms_per_step_without_exchange', 'measured_efficiency', 'predicted_efficiency_unoverlapped',
                                 'predicted_efficiency_fully_overlapped'}


def test_recurrence_timeout_falls_back_to_the_unoverlapped_schedule():
    """a watchdog timeout of a persistent recurrence kernel while the bucketed all-reduce runs beside it must not
    end the multi-GPU run: every rank raises in the same step (simulated here in the stub step), bench.py reruns with one
    all-reduce in optimizer_step and says so."""
    out = _run('--gpus', '2', PTMI_BENCH_FAKE_TIMEOUT='1')
    sched = out['rccl']['schedule']
    assert sched['used'] == 'no_overlap' and sched['probe_ms_per_step']['overlap'] is None
    assert sched['probe_ms_per_step']['no_overlap'] > 0 and any('timeout' in n for n in sched['notes'])
    assert out['rccl']['overlap_allreduce'] is False and out['value'] > 0


def test_c4_micro_steps_single_allreduce_per_optimizer_step():
    out = _run('--gpus', '2', '--config', 'c4', '--no-overlap-allreduce')
    assert out['config']['global_batch'] == 64 * 2 * 4 and out['rccl']['overlap_allreduce'] is False
    assert out['config']['frames_per_step'] == 2 * 4 * 64 * 503


def test_single_process_dry():
    out = _run()
    assert out['n_gpus'] == 1 and 'rccl' not in out
