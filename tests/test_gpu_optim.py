"""Flat-bucket optimizer kernels (csrc/optim.hip) against torch.optim.Adam + clip_grad_norm_ on the CPU
(the reference's optimizer step: padertorch/train/optimizer.py:27-42,79-90; trainer.py:512-532)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _registered_ops():
    import padertorch_amd.ops.library  # noqa: F401  (defines torch.ops.ptmi.*)



SHAPES = [(48, 257), (48,), (514, 24), (514,), (3, 5, 7), (1,), (2400, 12), (6,)]      # odd sizes: segments that start unaligned


def _models(seed=0):
    torch.manual_seed(seed)
    cpu = [torch.nn.Parameter(torch.randn(*s)) for s in SHAPES]
    gpu = [torch.nn.Parameter(p.detach().clone().cuda()) for p in cpu]
    return cpu, gpu


def _grads(step, scale):
    g = torch.Generator().manual_seed(100 + step)
    return [torch.randn(*s, generator=g) * scale for s in SHAPES]


def _native(gpu, **kw):
    from padertorch_amd.train.optimizer import Adam
    opt = Adam(**kw)
    opt.set_parameters(gpu)
    opt.use_flat_grads()
    return opt


@pytest.mark.parametrize('n', [1, 3, 4, 1023, 65536 + 5, 23_500_003])
def test_grad_norm_vs_fp64(n):
    torch.manual_seed(n)
    x = torch.randn(n, device='cuda') * 3
    got = torch.ops.ptmi.grad_norm(x)
    ref = torch.linalg.vector_norm(x.double(), 2)
    assert abs(float(got) - float(ref)) <= 2e-7 * float(ref) + 1e-30
    assert float(got) == float(torch.ops.ptmi.grad_norm(x))          # reproducible: fixed slices, fixed fold order


@pytest.mark.parametrize('clip,wd', [(1e10, 0.0), (0.5, 0.0), (0.5, 0.01)])
def test_adam_matches_torch_cpu(clip, wd):
    cpu, gpu = _models()
    ref = torch.optim.Adam(cpu, lr=1e-3, weight_decay=wd, foreach=False)
    opt = _native(gpu, gradient_clipping=clip, lr=1e-3, weight_decay=wd)
    assert opt._native_ok()
    for step in range(6):
        gs = _grads(step, 0.1 if step % 2 else 3.0)
        for p, q, g in zip(cpu, gpu, gs):
            p.grad = g.clone()
            q.grad.copy_(g)                         # views into the flat bucket
        n_ref = torch.nn.utils.clip_grad_norm_(cpu, clip)
        ref.step()
        n_got = opt.clip_grad()
        opt.step_and_zero_grad()
        assert abs(float(n_got) - float(n_ref)) <= 1e-6 * float(n_ref)
        assert float(opt.flat_grads.flat.abs().max()) == 0.0           # zeroed by the same kernel
        for p, q in zip(cpu, gpu):
            np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().numpy(), rtol=2e-6, atol=2e-7)
    for p, q in zip(cpu, gpu):
        st_r, st_g = ref.state[p], opt.optimizer.state[q]
        assert float(st_g['step']) == float(st_r['step']) == 6
        np.testing.assert_allclose(st_g['exp_avg'].cpu().numpy(), st_r['exp_avg'].numpy(), rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(st_g['exp_avg_sq'].cpu().numpy(), st_r['exp_avg_sq'].numpy(), rtol=2e-6, atol=1e-12)


def test_step_without_zero_and_without_clip_call():
    cpu, gpu = _models(1)
    ref = torch.optim.Adam(cpu, lr=2e-3, foreach=False)
    opt = _native(gpu, gradient_clipping=0.1, lr=2e-3)
    gs = _grads(0, 1.0)
    for p, q, g in zip(cpu, gpu, gs):
        p.grad = g.clone()
        q.grad.copy_(g)
    ref.step()                                      # no clip_grad() call: the reference does not clip then either
    opt.step()
    for p, q, g in zip(cpu, gpu, gs):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().numpy(), rtol=2e-6, atol=2e-7)
        np.testing.assert_array_equal(q.grad.cpu().numpy(), g.numpy())      # step() alone leaves the gradients


@pytest.mark.parametrize('how', ['found_inf', 'nan_norm', 'inf_loss'])
def test_skipped_update_leaves_parameters_but_zeroes(how):
    """The three ways an update is skipped on the device: torch's found_inf flag, a non-finite gradient norm, a non-finite
    value handed in by the Trainer (the sum of the step's losses)."""
    _, gpu = _models(2)
    opt = _native(gpu, gradient_clipping=1.0)
    for q, g in zip(gpu, _grads(0, 1.0)):
        q.grad.copy_(g)
    opt.clip_grad()
    opt.step_and_zero_grad()
    before = [q.detach().clone() for q in gpu]
    for q, g in zip(gpu, _grads(1, 1.0)):
        q.grad.copy_(g)
    if how == 'nan_norm':
        gpu[0].grad[0, 0] = float('nan')
    norm = opt.clip_grad()
    assert np.isfinite(float(norm)) == (how != 'nan_norm')
    if how == 'found_inf':
        opt.optimizer.found_inf = torch.ones((), device='cuda')
    if how == 'inf_loss':
        opt.skip_if_not_finite = torch.full((), float('inf'), device='cuda')
    opt.step_and_zero_grad()
    for q, b in zip(gpu, before):
        assert torch.equal(q.detach(), b)
    assert float(opt.flat_grads.flat.abs().max()) == 0.0
    assert all(float(opt.optimizer.state[q]['step']) == 1 for q in gpu)
    # and the next, clean step is applied
    opt.optimizer.found_inf = None
    for q, g in zip(gpu, _grads(2, 1.0)):
        q.grad.copy_(g)
    opt.clip_grad()
    opt.step_and_zero_grad()
    assert all(float(opt.optimizer.state[q]['step']) == 2 for q in gpu)
    assert not torch.equal(gpu[0].detach(), before[0])


def test_step_bumps_parameter_versions():
    """The kernel writes through raw pointers; caches keyed on Parameter._version (operand scales, stacked weights) rely on
    the bump."""
    _, gpu = _models(5)
    opt = _native(gpu, gradient_clipping=1.0)
    v0 = [q._version for q in gpu]
    for q, g in zip(gpu, _grads(0, 1.0)):
        q.grad.copy_(g)
    opt.clip_grad()
    opt.step_and_zero_grad()
    assert all(q._version > v for q, v in zip(gpu, v0))


def test_state_dict_round_trip_continues_identically():
    _, a = _models(3)
    _, b = _models(3)
    oa = _native(a, gradient_clipping=0.7)
    ob = _native(b, gradient_clipping=0.7)

    def run(opt, params, steps):
        for s in steps:
            for q, g in zip(params, _grads(s, 1.0)):
                q.grad.copy_(g)
            opt.clip_grad()
            opt.step_and_zero_grad()
    run(oa, a, range(5))
    run(ob, b, range(2))
    sd = copy.deepcopy(ob.state_dict())
    assert set(sd) == {'state', 'param_groups'} and set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}
    _, c = _models(3)
    for q, src in zip(c, b):
        q.data.copy_(src.data)
    oc = _native(c, gradient_clipping=0.7)
    oc.load_state_dict(sd)                          # torch replaces the state tensors: the next step re-binds them
    run(oc, c, range(2, 5))
    for q, r in zip(c, a):
        assert torch.equal(q.detach(), r.detach())


def test_torch_path_still_available():
    _, gpu = _models(4)
    opt = _native(gpu, gradient_clipping=1.0)
    opt.native = False
    for q, g in zip(gpu, _grads(0, 1.0)):
        q.grad.copy_(g)
    n = opt.clip_grad()
    opt.step_and_zero_grad()
    assert np.isfinite(float(n)) and float(opt.flat_grads.flat.abs().max()) == 0.0
