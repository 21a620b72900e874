"""``split_for_allreduce``: the captured optimizer step WITH a process group (``train.graphed.GraphedStep.split``; reference
``padertorch/train/trainer.py:396-442``) - graph A = every micro-step's forward + backward, the 'flat+words' exchange as ordinary RCCL
calls between the graphs, graph B = norm + clip + Adam.  One MI355X per test box: the process group has ONE rank (backend "nccl" = RCCL;
``all_reduce(SUM)`` is the identity), which exercises everything but the wire: the two-graph capture sharing one memory pool, the
collectives issued between two replays on the same stream, the update gate from the exchanged words, the Trainer's loop.  The
collective ORDER with more than one rank - also a rank without an example in a short last group - is covered on CPU
(``tests/test_trainer.py``, gloo, W = 2)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
LW = dict(pit_ips_loss=1., pit_mse_loss=0.)


@pytest.fixture
def one_rank_group():
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        yield dist
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def _pit(seed=0, **kw):
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    torch.manual_seed(seed)
    return PermutationInvariantTrainingModel(**dict(dict(F=257, recurrent_layers=2, units=32, K=2), **kw))


def _examples(n, B=4, N=6000, seed=0):
    from padertorch_amd.ops import pit_features
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        s = 0.1 * torch.randn(B, 2, N, generator=g)
        f = pit_features(s.sum(1).to(DEV), s.to(DEV))
        out.append({k: (list(v) if isinstance(v, list) else v) for k, v in f.items()})
    torch.cuda.synchronize()
    return out


def _train(model, exs, tmp, steps, vmb=2, **kw):
    import padertorch_amd as pt
    t = pt.Trainer(model, tmp, pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW, summary_trigger=(1, 'iteration'),
                   checkpoint_trigger=(1000, 'iteration'), stop_trigger=(steps, 'iteration'), virtual_minibatch_size=vmb, **kw)
    t.train(exs, device=DEV)
    return t


def test_trainer_graph_steps_with_a_process_group(tmp_path, one_rank_group):
    """``Trainer(graph_steps=True)`` under a process group: every optimizer step - the eager first sighting, the capture, the replays -
    issues exactly the protocol's two collectives (flat bucket, two words) in that order; parameters, losses and gradient norms equal
    those of the eager bucketed data-parallel loop and of the loop without a process group."""
    dist = one_rank_group
    exs = _examples(12)
    a, b = _pit(), _pit()
    ta = _train(a, exs, tmp_path / 'a', 6)                              # eager: layer buckets under the backward pass
    issued = []
    real = dist.all_reduce
    dist.all_reduce = lambda tensor, *args, **kw: (issued.append(tensor.numel()), real(tensor, *args, **kw))[1]
    try:
        tb = _train(b, exs, tmp_path / 'b', 6, graph_steps=True)
    finally:
        dist.all_reduce = real
    nflat = sum(p.numel() for p in b.parameters())
    assert issued == [nflat, 2] * 6, issued
    assert ta.iteration == tb.iteration == 6 and tb.dp_protocol is None and tb.deferred_checks is False
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        np.testing.assert_allclose(w.cpu().numpy(), v.cpu().numpy(), rtol=0, atol=1e-6, err_msg=k)
    sa = [s[2] for s in ta.summaries if s[1] == 'training']
    sb = [s[2] for s in tb.summaries if s[1] == 'training']
    assert len(sa) == len(sb) >= 5
    for x, y in zip(sa, sb):
        for key in x:
            np.testing.assert_allclose(y[key], x[key], rtol=1e-5, atol=1e-7, err_msg=key)


def test_split_step_non_finite_loss_raises_in_its_iteration(tmp_path, one_rank_group):
    """A NaN input in a REPLAYED data-parallel step: the summed loss word gates the update on the device, the error is the reference's,
    raised in the same iteration; the parameters are those after the last good step."""
    import padertorch_amd as pt
    exs = _examples(12)
    ref = _pit()
    _train(ref, exs, tmp_path / 'a', 4)
    bad = [dict(e) for e in exs]
    bad[8] = dict(bad[8], X_abs=[x * float('nan') for x in bad[8]['X_abs']])
    model = _pit()
    t = pt.Trainer(model, tmp_path / 'b', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW, summary_trigger=(1000, 'iteration'),
                   checkpoint_trigger=(1000, 'iteration'), stop_trigger=(6, 'iteration'), virtual_minibatch_size=2, graph_steps=True)
    with pytest.raises(RuntimeError, match='is not finite'):
        t.train(bad, device=DEV)
    assert t.iteration == 4, t.iteration
    for (k, v), (_, r) in zip(model.state_dict().items(), ref.state_dict().items()):
        np.testing.assert_allclose(v.cpu().numpy(), r.cpu().numpy(), rtol=0, atol=1e-6, err_msg=k)


@pytest.mark.parametrize('config', ['c2', 'c4'])
def test_split_step_at_baseline_size_equals_the_single_graph(tmp_path, one_rank_group, config):
    """BASELINE configs[1] (B = 32 x 4 s at 8 kHz) and configs[3]'s work of one GPU (B = 64 at 16 kHz, 4 micro-steps; 1 s signals): the
    two-graph step with the exchange between the graphs against the ONE-graph step of a twin model without the exchange - parameters
    bit-identical after three steps on changing batches, the GPU time of the three parts recorded."""
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    from padertorch_amd.ops import lstm as _lstm
    from padertorch_amd.train.graphed import GraphedStep
    B, fs, micro, n = (32, 8000, 1, 32000) if config == 'c2' else (64, 16000, 4, 16000)

    def waves(seed):
        g = torch.Generator().manual_seed(seed)
        s = 0.1 * torch.randn(B, 2, n, generator=g)
        return dict(y=s.sum(1).to(DEV), s=s.to(DEV), num_samples=[n] * B)

    def features(src):
        return pt.ops.pit_features(src['y'], src['s'], src['num_samples'])

    def make(path, dp):
        torch.manual_seed(5)
        m = PermutationInvariantTrainingModel()
        t = pt.Trainer(m, path, pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW, virtual_minibatch_size=micro, deferred_checks=True)
        t.to(torch.device(DEV))
        t._flat = t.optimizer.use_flat_grads()
        t.op_context.defer_wgrad = True
        _lstm.warm_side_stream(torch.device(DEV))
        m.train()
        if dp:
            t.dp_protocol = 'flat+words'
        else:
            t._dp_active = lambda: False            # (the twin: the same process, no exchange)
        return m, t
    steps = [[waves(100 * r + m) for m in range(micro)] for r in range(3)]
    results = []
    for dp in (False, True):
        m, t = make(tmp_path / str(dp), dp)
        static = [dict(y=w['y'].clone(), s=w['s'].clone(), num_samples=list(w['num_samples'])) for w in steps[0]]
        g = GraphedStep(t, static, prepare=features, warmup=1)
        assert g.split == dp
        g.times = [] if dp else None
        losses = []
        for batches in steps:
            g(batches)
            losses.append(g.scalars()['loss'])
        results.append(({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, losses, g.times))
        del g, m, t
    (pa, la, _), (pb, lb, times) = results
    assert la == lb, (la, lb)
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k
    assert len(times) == 3 and all(a > 0 and x >= 0 and b > 0 for a, x, b in times), times


# ------------------------------------------------------------------------------------------------------------------------------------
# TWO ranks for real: two processes share cuda:0, the collectives go through gloo (a one-GPU box cannot form an RCCL group of two).
def _two_rank_worker(rank, world, port, out_dir, graph, odd_on_rank1):
    import torch.distributed as dist
    import padertorch_amd as pt
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    try:
        exs = _examples(11, B=4, N=6000, seed=5)                 # 11 examples: five full groups of two and a short last one (rank 1 idle)
        if odd_on_rank1:
            # the example rank 1 takes in the fourth group has another length: rank 1 runs that step eagerly while rank 0 replays
            odd = _examples(1, B=4, N=6400, seed=9)[0]
            exs[7] = odd
        model = _pit(seed=3 + rank)                               # ranks start different: the step-0 broadcast fixes it
        issued = []
        real = dist.all_reduce
        dist.all_reduce = lambda tensor, *a, **k: (issued.append(tensor.numel()), real(tensor, *a, **k))[1]
        t = pt.Trainer(model, f'{out_dir}/r{rank}_{int(graph)}', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW, summary_trigger=(1000, 'iteration'),
                       checkpoint_trigger=(1000, 'iteration'), stop_trigger=(1, 'epoch'), virtual_minibatch_size=2, graph_steps=graph)
        made = []
        from padertorch_amd.train import graphed as G
        plain = G.GraphedStep.__init__
        G.GraphedStep.__init__ = lambda self, *a, **k: (made.append(1), plain(self, *a, **k))[1]
        t.train(exs, device=DEV)
        torch.cuda.synchronize()
        torch.save(dict(sd={k: v.detach().cpu() for k, v in model.state_dict().items()}, issued=issued, captures=len(made), iteration=t.iteration),
                   f'{out_dir}/out{rank}_{int(graph)}.pth')
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('odd_on_rank1', [False, True])
def test_two_ranks_train_with_captured_steps(tmp_path, odd_on_rank1):
    """W = 2 with replayed steps on BOTH ranks (gloo between two processes on one GPU): six optimizer steps - eager first sighting, capture,
    replays, a short last group with rank 1 idle; with ``odd_on_rank1`` one step in which rank 1 sees a new shape and runs eagerly while
    rank 0 replays.  Every rank issues [flat bucket, 2 words] per step and nothing else after the step-0 broadcast, replicas end
    bit-identical and equal the eager data-parallel run's parameters."""
    import torch.multiprocessing as mp
    out = {}
    for graph in (False, True):
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), graph, odd_on_rank1), nprocs=2, join=True)
        out[graph] = [torch.load(tmp_path / f'out{r}_{int(graph)}.pth') for r in range(2)]
    g0, g1 = out[True]
    assert g0['iteration'] == g1['iteration'] == 6
    for k in g0['sd']:
        assert torch.equal(g0['sd'][k], g1['sd'][k]), k                  # replicas identical
        np.testing.assert_allclose(g0['sd'][k].numpy(), out[False][0]['sd'][k].numpy(), rtol=0, atol=2e-6, err_msg=k)
    nflat = sum(v.numel() for k, v in g0['sd'].items())
    for r in (g0, g1):
        big = [n for n in r['issued'] if n > 1]
        assert big == [nflat, 2] * 6, big
        assert r['captures'] >= 1
    assert g0['captures'] == 1 and g1['captures'] == (1 if not odd_on_rank1 else 1)


def test_opt_in_captured_exchange_with_a_one_rank_group(tmp_path, one_rank_group):
    """``Trainer.graph_exchange = 'captured'`` (opt-in): ONE graph per optimizer step whose nodes include the layer buckets' RCCL
    all-reduces on the weight-gradient queue and the update gate's - RCCL collectives do capture into a hipGraph
    (``scripts/mb/rccl_in_graph.py``).  A one-rank group is all a one-GPU box can form: same parameters and losses as the eager
    data-parallel loop; no collective is issued outside the replays."""
    dist = one_rank_group
    exs = _examples(12)
    a, b = _pit(), _pit()
    ta = _train(a, exs, tmp_path / 'a', 6)
    import padertorch_amd as pt
    tb = pt.Trainer(b, tmp_path / 'b', pt.optimizer.Adam(gradient_clipping=1.), loss_weights=LW, summary_trigger=(1, 'iteration'),
                    checkpoint_trigger=(1000, 'iteration'), stop_trigger=(6, 'iteration'), virtual_minibatch_size=2, graph_steps=True)
    tb.graph_exchange = 'captured'
    issued = []
    real = dist.all_reduce
    dist.all_reduce = lambda tensor, *args, **kw: (issued.append(tensor.numel()), real(tensor, *args, **kw))[1]
    try:
        tb.train(exs, device=DEV)
    finally:
        dist.all_reduce = real
    assert ta.iteration == tb.iteration == 6
    # python saw the collectives of the eager first step and of the capture (which executes nothing); the four replays issue theirs as nodes
    per_step = len([n for n in issued]) // 2
    assert len(issued) == 2 * per_step and per_step >= 4, issued
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        np.testing.assert_allclose(w.cpu().numpy(), v.cpu().numpy(), rtol=0, atol=1e-6, err_msg=k)
    sa = [s[2] for s in ta.summaries if s[1] == 'training']
    sb = [s[2] for s in tb.summaries if s[1] == 'training']
    for x, y in zip(sa, sb):
        for key in x:
            np.testing.assert_allclose(y[key], x[key], rtol=1e-5, atol=1e-7, err_msg=key)
