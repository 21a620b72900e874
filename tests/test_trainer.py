"""Host logic of padertorch_amd.Trainer on CPU: loop semantics vs the reference Trainer (G6),
checkpoints/resume, error handling, and the data-parallel path with gloo (world_size 2)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch

import padertorch_amd as pt
from oracle import torch_ref


class RefModel(torch_ref.PITModelRef):
    """The oracle's torch-CPU model behind the Model API the Trainer touches."""
    create_snapshot = False

    def example_to_device(self, example, device=None, memo=None):
        return pt.data.example_to_device(example, device, memo)

    def modify_summary(self, summary):
        return summary


def _batch(g6):
    Ts = [int(t) for t in g6['Ts']]
    b = {k: [g6[f'in_{k}_{i}'] for i in range(len(Ts))]          # numpy: example_to_device converts
         for k in ['Y_abs', 'X_abs', 'cos_phase_difference']}
    b['num_frames'] = Ts
    return b


def _model(g6):
    m = RefModel(F=9, recurrent_layers=2, units=4, K=2)
    m.load_state_dict({k[len('pit_sd_'):]: torch.from_numpy(v) for k, v in g6.items()
                       if isinstance(v, np.ndarray) and k.startswith('pit_sd_')})
    return m


def _examples(g6):
    batch = _batch(g6)
    return [{k: [v[b] for b in idx] for k, v in batch.items()} for idx in g6['train_example_indices']]


LW = dict(pit_ips_loss=1., pit_mse_loss=0.)


def test_three_steps_match_reference_trainer(g6, tmp_path):
    """Adam(clip=1), virtual_minibatch_size=2, 6 examples -> the reference Trainer's parameters."""
    model = _model(g6)
    trainer = pt.Trainer(model, tmp_path, pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW,
                         summary_trigger=(1000, 'iteration'), checkpoint_trigger=(1000, 'iteration'),
                         stop_trigger=(3, 'iteration'), virtual_minibatch_size=2)
    trainer.train(_examples(g6), device='cpu')
    assert trainer.iteration == 3
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g6['pit_sd3_' + k], atol=1e-6, err_msg=k)
    # storage layout: ckpt_0 (first pre_step), ckpt_3 (close), ckpt_latest -> ckpt_3
    files = sorted(p.name for p in (tmp_path / 'checkpoints').iterdir())
    assert files == ['ckpt_0.pth', 'ckpt_3.pth', 'ckpt_latest.pth'], files
    assert os.readlink(tmp_path / 'checkpoints' / 'ckpt_latest.pth') == 'ckpt_3.pth'
    # zero-weight losses are logged but not part of the objective (trainer.py:608-613)
    sc = trainer.summaries[-1][2]
    assert {'pit_mse_loss', 'pit_ips_loss', 'loss', 'grad_norm', 'lr/param_group_0'} <= set(sc)


def test_resume_continues_bit_exact(g6, tmp_path):
    exs = _examples(g6)
    kw = dict(loss_weights=LW, summary_trigger=(1000, 'iteration'), checkpoint_trigger=(1, 'iteration'),
              virtual_minibatch_size=1)
    a = pt.Trainer(_model(g6), tmp_path / 'a', pt.optimizer.Adam(1.), stop_trigger=(4, 'iteration'), **kw)
    a.train(exs, device='cpu')
    b = pt.Trainer(_model(g6), tmp_path / 'b', pt.optimizer.Adam(1.), stop_trigger=(2, 'iteration'), **kw)
    b.train(exs, device='cpu')
    b2 = pt.Trainer(_model(g6), tmp_path / 'b', pt.optimizer.Adam(1.), stop_trigger=(4, 'iteration'), **kw)
    b2.train(exs[2:], resume=True, device='cpu')
    assert b2.iteration == 4
    for (k, va), vb in zip(a.model.state_dict().items(), b2.model.state_dict().values()):
        assert torch.equal(va, vb), k


def test_non_finite_loss_raises_and_dumps(g6, tmp_path):
    exs = _examples(g6)
    exs[0]['Y_abs'][0] = exs[0]['Y_abs'][0].copy()      # the fixture is session scoped
    exs[0]['Y_abs'][0][0, 0] = np.nan
    t = pt.Trainer(_model(g6), tmp_path, pt.optimizer.Adam(1.), loss_weights=LW)
    with pytest.raises(RuntimeError, match='not finite'):
        t.train(exs, device='cpu')
    assert list((tmp_path / 'log').glob('error_state_*'))


def test_review_key_and_loss_weight_checks(g6, tmp_path):
    t = pt.Trainer(_model(g6), tmp_path, pt.optimizer.Adam(1.), loss_weights=None)
    with pytest.raises(Exception, match='multiple losses'):
        t.train(_examples(g6), device='cpu')


def test_test_run_invariants(g6, tmp_path):
    model = _model(g6)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    t = pt.Trainer(model, tmp_path, pt.optimizer.Adam(1.), loss_weights=LW, virtual_minibatch_size=2)
    t.test_run(_examples(g6), _examples(g6)[:2], device='cpu')
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k            # parameters restored bit-exact
    assert not (tmp_path / 'checkpoints').exists()     # the run happened in temp dirs


# ------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, g6_path, n_examples, out_dir, overlap=True, poison_rank=None, protocol=None):
    import json
    torch.set_num_threads(1)
    g6 = dict(np.load(g6_path, allow_pickle=False))
    g6['train_example_indices'] = json.loads(str(g6['train_example_indices']))
    torch.distributed.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}',
                                         rank=rank, world_size=world)
    model = _model(g6)
    if rank == 1:                         # ranks start different: step-0 broadcast must fix it
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.)
    exs = (_examples(g6) * 2)[:n_examples]
    if poison_rank is not None:           # the example rank `poison_rank` sees in the second group carries a NaN
        ex = exs[2 + poison_rank]
        ex['Y_abs'] = [a.copy() for a in ex['Y_abs']]
        ex['Y_abs'][0][0, 0] = np.nan
    t = pt.Trainer(model, os.path.join(out_dir, f'r{rank}'), pt.optimizer.Adam(gradient_clipping=1.),
                   loss_weights=LW, summary_trigger=(1000, 'iteration'),
                   checkpoint_trigger=(1000, 'iteration'), stop_trigger=(1, 'epoch'),
                   virtual_minibatch_size=2, overlap_allreduce=overlap)
    t.dp_protocol = protocol
    issued = []
    real = torch.distributed.all_reduce
    torch.distributed.all_reduce = lambda tensor, *a, **k: (issued.append(tensor.numel()), real(tensor, *a, **k))[1]
    if poison_rank is None:
        t.train(exs, device='cpu')
        torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(out_dir, f'sd{rank}.pth'))
        torch.save(issued, os.path.join(out_dir, f'issued{rank}.pth'))
    else:
        try:
            t.train(exs, device='cpu')
            outcome = 'finished'
        except RuntimeError as e:
            outcome = str(e).split('\n')[0]
        with open(os.path.join(out_dir, f'outcome{rank}.txt'), 'w') as f:
            f.write(outcome)
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('n_examples,overlap', [(8, True), (7, True), (7, False)])
def test_data_parallel_gloo_matches_single_process(g6, tmp_path, n_examples, overlap):
    """W=2 ranks x 1 micro-step == 1 process x virtual_minibatch_size=2: gradients are SUMMED (no
    averaging), replicas end bit-identical, and a partial last group (7 examples) works."""
    import torch.multiprocessing as mp
    from conftest import GOLDEN
    port = _free_port()
    mp.spawn(_dp_worker, args=(2, port, str(GOLDEN / 'g6_models.npz'), n_examples, str(tmp_path), overlap),
             nprocs=2, join=True)
    sd0 = torch.load(tmp_path / 'sd0.pth')
    sd1 = torch.load(tmp_path / 'sd1.pth')
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k           # replicas identical
    model = _model(g6)
    exs = (_examples(g6) * 2)[:n_examples]
    t = pt.Trainer(model, tmp_path / 'single', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW,
                   summary_trigger=(1000, 'iteration'), checkpoint_trigger=(1000, 'iteration'),
                   stop_trigger=(1, 'epoch'), virtual_minibatch_size=2)
    t.train(exs, device='cpu')
    assert t.iteration == 4
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(sd0[k].numpy(), v.numpy(), atol=2e-6, err_msg=k)


@pytest.mark.parametrize('n_examples', [8, 7])
def test_data_parallel_flat_and_words_protocol(g6, tmp_path, n_examples):
    """The protocol of captured data-parallel steps (``Trainer.dp_protocol = 'flat+words'``, what ``graph_steps`` switches on under a
    process group; the graphs themselves need a GPU: ``tests/test_gpu_graphed_dp.py``) on its eager path, W = 2: per optimizer step
    every rank issues all_reduce(flat bucket) then all_reduce(2 words) - also the rank WITHOUT an example in the short last group of 7
    -, same order on both ranks, replicas bit-identical and equal to the single process."""
    import torch.multiprocessing as mp
    from conftest import GOLDEN
    mp.spawn(_dp_worker, args=(2, _free_port(), str(GOLDEN / 'g6_models.npz'), n_examples, str(tmp_path), True, None, 'flat+words'),
             nprocs=2, join=True)
    sd0, sd1 = torch.load(tmp_path / 'sd0.pth'), torch.load(tmp_path / 'sd1.pth')
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k
    issued = [torch.load(tmp_path / f'issued{r}.pth') for r in range(2)]
    nflat = sum(v.numel() for v in _model(g6).parameters())
    assert issued[0] == issued[1]
    big = [n for n in issued[0] if n > 1]
    assert big == [nflat, 2] * 4, issued[0]                   # four optimizer steps; (the 1-element entries: the sync loss checks)
    model = _model(g6)
    exs = (_examples(g6) * 2)[:n_examples]
    t = pt.Trainer(model, tmp_path / 'single', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW,
                   summary_trigger=(1000, 'iteration'), checkpoint_trigger=(1000, 'iteration'),
                   stop_trigger=(1, 'epoch'), virtual_minibatch_size=2)
    t.train(exs, device='cpu')
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(sd0[k].numpy(), v.numpy(), atol=2e-6, err_msg=k)


def test_data_parallel_non_finite_loss_raises_on_every_rank(g6, tmp_path):
    """Sync checks, W = 2: a NaN loss on ONE rank makes BOTH raise (nobody is left blocking in the all-reduce)."""
    import torch.multiprocessing as mp
    from conftest import GOLDEN
    mp.spawn(_dp_worker, args=(2, _free_port(), str(GOLDEN / 'g6_models.npz'), 8, str(tmp_path), True, 1),
             nprocs=2, join=True)
    out = [(tmp_path / f'outcome{r}.txt').read_text() for r in range(2)]
    assert 'not finite' in out[1] and 'not finite' in out[0], out


def _shared_module_worker(rank, world, port, out_dir):
    """A module applied twice in one forward pass whose weight gradient is accumulated IN PLACE (the ops.lstm / ops.linear route,
    imitated here on CPU): the forward pass announces every use (GradBuckets.expect through ops.lstm.GRAD_USE_HOOK), the backward
    pass reports every finished use; the bucket's all-reduce must start only after the LAST one."""
    from padertorch_amd.ops import lstm as _lstm
    from padertorch_amd.train.trainer import GradBuckets
    torch.set_num_threads(1)
    torch.distributed.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 6), torch.nn.Linear(6, 3))
    shared = model[0]
    opt = pt.optimizer.Adam(1.)
    opt.set_parameters(model.parameters())
    flat = opt.use_flat_grads()
    buckets = GradBuckets(model, flat)
    assert [n for _, _, n in buckets.buckets] == [2, 2]
    _lstm.GRAD_USE_HOOK = buckets.expect
    _lstm.GRAD_READY_HOOK = lambda params: buckets.ready(params, None)
    issued_at = []

    class InPlaceLinear(torch.autograd.Function):          # dW accumulated into .grad by the op itself, like ops.linear under DEFER_WGRAD
        @staticmethod
        def forward(ctx, x, mod):
            ctx.mod = mod
            ctx.save_for_backward(x)
            return x @ mod.weight.detach().t() + mod.bias.detach()

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            mod = ctx.mod
            before = len(buckets.works)
            mod.weight.grad.add_(g.t() @ x)
            mod.bias.grad.add_(g.sum(0))
            _lstm.GRAD_READY_HOOK([mod.weight, mod.bias])
            issued_at.append(len(buckets.works) - before)
            return g @ mod.weight.detach(), None

    def apply(mod, x):
        _lstm.GRAD_USE_HOOK([mod.weight, mod.bias])
        return InPlaceLinear.apply(x, mod)

    try:
        buckets.active = True
        x = torch.full((4, 6), float(rank + 1), requires_grad=True)
        y = apply(model[1], apply(shared, torch.tanh(apply(shared, x))))          # shared module: two uses
        y.sum().backward()
        # backward order: model[1] (bucket 1 issued), shared 2nd use (nothing: one use still to come), shared 1st use (bucket 0 issued)
        assert issued_at == [1, 0, 1], issued_at
        buckets.finish()
        torch.save(flat.flat.clone(), os.path.join(out_dir, f'flat{rank}.pth'))
        # reference: the same two ranks' gradients through plain autograd, summed
        want = torch.zeros_like(flat.flat)
        for r in range(world):
            ref = torch.nn.Sequential(torch.nn.Linear(6, 6), torch.nn.Linear(6, 3))
            ref.load_state_dict(model.state_dict())
            xr = torch.full((4, 6), float(r + 1))
            ref[1](ref[0](torch.tanh(ref[0](xr)))).sum().backward()
            want += torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
        torch.testing.assert_close(flat.flat, want, atol=1e-5, rtol=1e-5)
    finally:
        _lstm.GRAD_USE_HOOK = None
        _lstm.GRAD_READY_HOOK = None
        torch.distributed.destroy_process_group()


def test_shared_module_bucket_waits_for_its_last_backward_use(tmp_path):
    """a module applied twice must not have its bucket reduced after the first backward use (W = 2, gloo)."""
    import torch.multiprocessing as mp
    mp.spawn(_shared_module_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / 'flat0.pth'), torch.load(tmp_path / 'flat1.pth')
    assert torch.equal(a, b)


def test_grad_buckets_follow_layers_and_issue_last_first(g6):
    from padertorch_amd.train.trainer import GradBuckets
    model = _model(g6)
    opt = pt.optimizer.Adam(1.)
    opt.set_parameters(model.parameters())
    flat = opt.use_flat_grads()
    b = GradBuckets(model, flat)
    # 2 BLSTM layers (8 parameters each) + linear1 + linear2, contiguous and complete
    assert [n for _, _, n in b.buckets] == [8, 8, 2, 2]
    assert b.buckets[0][0] == 0 and b.buckets[-1][1] == flat.flat.numel()
    assert all(b.buckets[i][1] == b.buckets[i + 1][0] for i in range(3))
    assert b.next == 3 and not b.active


def test_resume_restores_best_and_does_not_refire_triggers(g6, tmp_path):
    exs = _examples(g6)
    kw = dict(loss_weights=LW, summary_trigger=(1000, 'iteration'), checkpoint_trigger=(1, 'iteration'), virtual_minibatch_size=1)
    a = pt.Trainer(_model(g6), tmp_path, pt.optimizer.Adam(1.), stop_trigger=(2, 'iteration'), **kw)
    a.register_validation_hook(exs[:1])
    a.train(exs, device='cpu')
    best = a._best
    assert best is not None
    ckpt = torch.load(tmp_path / 'checkpoints' / 'ckpt_latest.pth', weights_only=False)
    assert ckpt['hooks'] == {} and ckpt['ptmi_hooks']['best'] == best       # 'hooks' stays loadable by the reference Trainer
    mtime = (tmp_path / 'checkpoints' / 'ckpt_2.pth').stat().st_mtime_ns
    b = pt.Trainer(_model(g6), tmp_path, pt.optimizer.Adam(1.), stop_trigger=(2, 'iteration'), **kw)
    b.register_validation_hook(exs[:1])
    b.train(exs, resume=True, device='cpu')            # already at the stop iteration: nothing to do
    assert b._best == best
    assert (tmp_path / 'checkpoints' / 'ckpt_2.pth').stat().st_mtime_ns == mtime      # not validated / saved again


def test_from_storage_dir_loads_a_reference_style_storage_dir(tmp_path):
    """``Module.from_storage_dir`` (reference ``base.py:183-225``): the config file the reference Trainer writes names the
    REFERENCE class; it resolves to this package's class, nested factories / partials are built, the weights come from
    ``checkpoints/ckpt_best_loss.pth['model']``."""
    import json
    import torch
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    torch.manual_seed(0)
    src = PermutationInvariantTrainingModel(F=9, recurrent_layers=2, units=4, K=3)
    config = {'trainer': {'factory': 'padertorch.train.trainer.Trainer',
                          'model': {'factory': 'padertorch.contrib.examples.source_separation.pit.model.'
                                               'PermutationInvariantTrainingModel',
                                    'F': 9, 'recurrent_layers': 2, 'units': 4, 'K': 3, 'dropout_input': 0.0,
                                    'dropout_hidden': 0.0, 'dropout_linear': 0.0, 'output_activation': 'relu'},
                          'optimizer': {'factory': 'padertorch.train.optimizer.Adam', 'lr': 0.001}}}
    (tmp_path / 'checkpoints').mkdir()
    (tmp_path / 'config.json').write_text(json.dumps(config))
    torch.save({'model': src.state_dict(), 'iteration': 7}, tmp_path / 'checkpoints' / 'ckpt_best_loss.pth')
    got = pt.Module.from_storage_dir(tmp_path)
    assert type(got) is PermutationInvariantTrainingModel and got.K == 3
    for (k, a), b in zip(src.state_dict().items(), got.state_dict().values()):
        assert torch.equal(a, b), k
    # an inner module, a partial and a plain torch factory
    inner = {'net': {'factory': 'torch.nn.Linear', 'in_features': 3, 'out_features': 2},
             'act': {'partial': 'torch.nn.functional.leaky_relu', 'negative_slope': 0.5}}
    from padertorch_amd.base import _instantiate
    built = _instantiate(inner)
    assert isinstance(built['net'], torch.nn.Linear) and float(built['act'](torch.tensor(-2.0))) == -1.0
    opt = _instantiate(config['trainer']['optimizer'])
    assert isinstance(opt, pt.optimizer.Adam)


def test_resume_keeps_the_validation_metric_of_the_current_run(g6, tmp_path):
    """load_state_dict must not overwrite the metric / direction registered by the resuming run; the old metric's best
    value does not carry over to a different metric."""
    exs = _examples(g6)
    kw = dict(loss_weights=LW, summary_trigger=(1000, 'iteration'), checkpoint_trigger=(1, 'iteration'), virtual_minibatch_size=1)
    a = pt.Trainer(_model(g6), tmp_path, pt.optimizer.Adam(1.), stop_trigger=(1, 'iteration'), **kw)
    a.register_validation_hook(exs[:1])
    a.train(exs, device='cpu')
    assert a._best is not None
    b = pt.Trainer(_model(g6), tmp_path, pt.optimizer.Adam(1.), stop_trigger=(1, 'iteration'), **kw)
    b.register_validation_hook(exs[:1], metric='pit_mse_loss', maximize=True)
    with pytest.warns(UserWarning, match='starting a new best'):
        b.load_checkpoint()
    assert b.validation_metric == 'pit_mse_loss' and b.validation_maximize is True and b._best is None


def test_weight_caches_do_not_survive_their_parameter(monkeypatch):
    """the operand-scale / plane caches are keyed by id(parameter); an entry must not match another parameter object
    that happens to reuse the id, the version and the storage pointer (a freed model followed by a freshly loaded one)."""
    from padertorch_amd.ops import gemm as G
    calls = []
    monkeypatch.setattr(G, 'absmax', lambda x: calls.append(1) or x.abs().max().reshape(1))
    G.invalidate()
    p1 = torch.nn.Parameter(torch.ones(3, 4))
    v1 = G.weight_absmax(p1)
    assert G.weight_absmax(p1) is v1 and len(calls) == 1              # a hit for the very same object
    p2 = torch.nn.Parameter(torch.full((3, 4), 2.))
    entry = G._WEIGHT_AMAX[id(p1)]
    G._WEIGHT_AMAX[id(p2)] = (p2._version, p2.data_ptr(), entry[2], entry[3], None)      # the coincidence: p1's entry under p2's identity
    assert float(G.weight_absmax(p2)) == 2. and len(calls) == 2         # not served from p1's entry
    del p1
    import gc
    gc.collect()
    assert entry[3][0]() is None                                         # the entry held no strong reference
    G.invalidate()
    assert not G._WEIGHT_AMAX and not G._WEIGHT_PLANES
