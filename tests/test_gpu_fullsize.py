"""Whole-step oracle parity at BASELINE size (VERDICT r1, weak point 1).

The model-level goldens are toy nets; the full-size kernels were only covered through self-consistency.  Here ONE
training step (forward, review, backward: masks, losses, every parameter gradient) of the real configurations runs on the
GPU and on the CPU oracle (``oracle/torch_ref.py``: torch.nn.LSTM / Linear / the reference's python-loop losses, fp32) from
the same weights and the same input features:

  * BASELINE configs[1]: PIT, 3 x BLSTM-600, B = 32, T = 253 (8 kHz): the bench workload, 16-row chains;
  * configs[2] shape:    PIT, B = 40 of the 64, T = 503 (16 kHz): more than 32 rows -> the 32-row-chain kernels, 503 steps;
  * configs[4] shape:    deep clustering, 2 x BLSTM-600, E = 20, K = 3, B = 34, T = 503, a ragged tail (shorter examples).

A hand-off race in the persistent recurrence (a stale tile, a missed flag) would show up here as a wrong mask; tolerances:
masks / embeddings atol 1e-5, losses 1e-4 (BASELINE.json north_star) against the fp32 oracle; gradients within 2e-4 of each parameter's largest gradient
entry against the SAME oracle step in fp64 (``_grad_check``; round 6: the double run goes through torch's native kernels on the GPU, so every case has it).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _waveforms(B, K, n, lens, seed):
    g = torch.Generator().manual_seed(seed)
    s = 0.1 * torch.randn(B, K, n, generator=g)
    for b, l in enumerate(lens):
        s[b, :, l:] = 0.
    return s


def _double_oracle(ref, run):
    """The same oracle step in fp64 (``run(model64, to64)`` does forward, review and backward): what both fp32 gradients - the HIP
    path's and the CPU oracle's - are measured against."""
    import copy
    # (on the GPU, through torch's native double kernels - MIOpen has no double LSTM -: a second instead of minutes on the CPU; test
    #  infrastructure, nothing of libptmi.so runs here)
    ref64 = copy.deepcopy(ref).double().to(DEV)
    for p in ref64.parameters():
        p.grad = None
    run(ref64, lambda t: t.detach().double().to(DEV))
    return ref64.cpu()


def _grad_check(model, ref, ref64=None, tol=2e-4):
    """Every parameter gradient of the HIP step against the fp64 oracle (the fp32 CPU oracle where a case skips the double run): within
    ``tol`` of the gradient's largest entry.  With the fp64 run both fp32 gradients - the HIP path's and the CPU oracle's - are measured
    against the same truth (``PTMI_GRAD_REPORT=<file>`` appends them per parameter: ``profiles/r5_grad_errors_vs_fp64.txt``): over all
    other cases the HIP path's error is 1e-6 in the median and 4.8e-5 at worst, the CPU's 3e-7 / 1.0e-4.  Two row-slot cases keep a 5e-4
    gate: B = 100 in 64 slots (3 x BLSTM-64; ``linear1.weight`` off by 3.3e-4) and B = 70 in 32 static slots (2 x BLSTM-600; 2.8e-4).
    Round 5 blamed the 16-bit planes of the gradient operand; round 6 measured it (``scripts/dbg_wgrad.py``,
    ``profiles/r6_relu_tie.txt``): the weight-gradient GEMMs are exact to 2e-7 on their own inputs, and in both cases ONE ReLU of
    ``linear1`` whose pre-activation is 1e-10 / 1e-9 takes the other branch than in the fp64 chain - with the HIP path's own ReLU
    pattern the whole fp64 chain reproduces the HIP gradient to 2e-7.  A subgradient tie, not a precision loss: the gate of those
    cases covers one such flip."""
    worst = {}
    truth = ref64 if ref64 is not None else ref
    report = []
    for (n, p), (_, q), (_, q64) in zip(model.named_parameters(), ref.named_parameters(), truth.named_parameters()):
        g, r32, r = p.grad.detach().cpu().double(), q.grad.double(), q64.grad.double()
        scale = float(r.abs().max())
        err = float((g - r).abs().max())
        err_cpu = float((r32 - r).abs().max())
        worst[n] = err / max(scale, 1e-30)
        report.append((n, err / max(scale, 1e-30), err_cpu / max(scale, 1e-30)))
    import os
    if os.environ.get('PTMI_GRAD_REPORT') and ref64 is not None:
        with open(os.environ['PTMI_GRAD_REPORT'], 'a') as f:
            for n, e, c in report:
                f.write(f'{n} hip {e:.3e} cpu32 {c:.3e}\n')
            f.write('--\n')
    for n, e, c in report:
        assert e <= tol, (n, 'HIP error / largest entry', e, 'fp32 CPU oracle', c)
    return worst


def _pit_case(B, fs, lens=None, seed=0, n=None, row_slots=None, grad_tol=2e-4, in_place=False, double=True, static_slots=None,
              device_lengths=False, **model_kw):
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    from padertorch_amd.ops import lstm as _lstm
    from oracle import torch_ref
    n = n or 4 * fs
    lens = lens or [n] * B
    torch.manual_seed(seed)
    model = PermutationInvariantTrainingModel(**model_kw)
    ref = torch_ref.PITModelRef(**model_kw)
    ref.load_state_dict(model.state_dict())
    model.to(DEV).train()
    model.row_slots = row_slots
    s = _waveforms(B, model_kw.get('K', 2), n, lens, seed + 1).to(DEV)
    feats = pt.ops.pit_features(s.sum(1), s, lens)
    if static_slots is not None:
        # a row-slot layout of FIXED capacity whose pattern is device data (ops.sequence.StaticSlots): (slots, spare steps)
        from padertorch_amd.ops.sequence import SlotLayout, StaticSlots
        frames = list(feats['num_frames'])
        need = SlotLayout(frames, static_slots[0]).T
        slots = StaticSlots(B, static_slots[0], need + static_slots[1], max(frames), DEV).set(frames)
        if device_lengths:      # ... and the features made from device-side lengths too (the launch knows the padded shape only)
            ns = torch.tensor(lens, dtype=torch.int32, device=DEV)
            dev_feats = pt.ops.pit_features(s.sum(1), s, ns, num_frames_dev=slots.frames)
            for k in ('Y_abs', 'X_abs', 'cos_phase_difference'):
                assert torch.equal(dev_feats[k].padded, feats[k].padded), k
            true_feats, feats = feats, dev_feats
        feats = dict(feats, slots=slots)
    _lstm.CHECK_PERSISTENT_ERRORS = True
    if in_place:        # the Trainer's route: weight gradients accumulated into existing .grad buffers on the side stream (ops.context)
        from padertorch_amd.ops import context as _context
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        _context.attach(model, _context.OpContext(defer_wgrad=True))
        _lstm.warm_side_stream(torch.device(DEV))
    try:
        masks = model(feats)
        losses = model.review(feats, masks)['losses']
        losses['pit_ips_loss'].backward()
        _lstm.sync_deferred()
    finally:
        _lstm.CHECK_PERSISTENT_ERRORS = False
    torch.cuda.synchronize()
    if static_slots is not None:
        if device_lengths:
            feats = true_feats
        for m, t in zip(masks, feats['num_frames']):
            assert float(m[t:].abs().sum()) == 0.           # padding frames of the padded mask tensor are zero
        masks = [m[:t] for m, t in zip(masks, feats['num_frames'])]
    # oracle: same features (copied), same weights
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    rb = {k: [t.detach().cpu() for t in feats[k]] for k in ('Y_abs', 'X_abs', 'cos_phase_difference')}
    rmasks = ref(rb)
    rlosses = ref.review(rb, rmasks)['losses']
    rlosses['pit_ips_loss'].backward()
    worst_mask = max(float((m.detach().cpu() - r.detach()).abs().max()) for m, r in zip(masks, rmasks))
    assert worst_mask < 1e-5, worst_mask
    for k in ('pit_mse_loss', 'pit_ips_loss'):
        assert abs(float(losses[k]) - float(rlosses[k])) < 1e-4, (k, float(losses[k]), float(rlosses[k]))

    def run64(m64, to64):
        b64 = {k: [to64(t) for t in feats[k]] for k in ('Y_abs', 'X_abs', 'cos_phase_difference')}
        m64.review(b64, m64(b64))['losses']['pit_ips_loss'].backward()
    return worst_mask, _grad_check(model, ref, _double_oracle(ref, run64) if double else None, grad_tol)


def test_pit_step_config2_size_vs_oracle():
    _pit_case(32, 8000)


@pytest.mark.parametrize('seed', range(16))
def test_pit_step_random_small_configurations_vs_oracle(seed):
    """Random model sizes (units, layers, K, output activation), batch sizes 1 ... 40 and ragged / equal lengths of 0.1 ... 0.6 s: the
    whole step (packed feature output, recurrences of every tile shape, dense layers, unpack scatter, PIT loss, every gradient)
    against the oracle - the corners the three BASELINE shapes do not touch."""
    rng = np.random.RandomState(1000 + seed)
    B = int(rng.randint(1, 41))
    n = int(rng.randint(800, 4800))
    ragged = bool(rng.randint(0, 2))
    lens = sorted((int(x) for x in rng.randint(n // 3, n + 1, B)), reverse=True) if ragged else [n] * B
    lens[0] = n
    _pit_case(B, 8000, lens=lens, seed=seed, n=n, units=int(rng.choice([4, 24, 100, 600])), recurrent_layers=int(rng.randint(1, 4)),
              K=int(rng.randint(2, 4)), output_activation=str(rng.choice(['relu', 'sigmoid'])))


@pytest.mark.parametrize('in_place', [False, True])
@pytest.mark.parametrize('B,slots,units,layers', [(10, 4, 24, 2), (40, 16, 100, 1), (70, 32, 600, 2), (100, 64, 64, 3), (33, 32, 600, 1),
                                                  (5, 1, 8, 2)])
def test_pit_step_on_row_slots_vs_oracle(B, slots, units, layers, in_place):
    """Ragged batches on the row-slot layout (model.row_slots; ops.sequence.SlotLayout): 2 - 5 sequences lie end to end in every
    row slot, the recurrences reset (h, c) at the boundaries in both directions (ptmi_lstm_forward / backward_persistent_slots), idle
    slot steps contribute nothing.  The whole step - masks of every example, both losses, every parameter gradient - against the
    oracle, which knows nothing of slots (one sequence per row, torch.nn.LSTM on the PackedSequence).  in_place: the Trainer's route
    (weight gradients accumulated in place on the side stream); with 16 / 32 / 64 slots the recurrences' planes then serve the
    projections, the input gradients and - as dgates^T - the weight gradients, as for equal-length batches."""
    from padertorch_amd.ops.sequence import SlotLayout
    rng = np.random.RandomState(B + slots)
    n = 4400
    lens = sorted((int(x) for x in rng.randint(900, n + 1, B)), reverse=True)
    lens[0] = n
    frames = [(v + 2 * 384 - 512 + 127) // 128 + 1 for v in lens]
    layout = SlotLayout(frames, slots)
    per_slot = np.bincount(layout.slot, minlength=slots)
    assert per_slot.max() >= 2 and layout.T == max(np.bincount(layout.slot, weights=frames, minlength=slots)), (per_slot, layout.T)
    # (B = 100: linear1.weight's gradient is off by 3.3e-4 of its largest entry against the fp64 oracle: ONE ReLU tie at a pre-activation
    #  of 1e-9, see _grad_check and profiles/r6_relu_tie.txt)
    _pit_case(B, 8000, lens=lens, seed=B, n=n, row_slots=slots, grad_tol=5e-4 if B == 100 else 2e-4, in_place=in_place, units=units,
              recurrent_layers=layers, K=2 + B % 2)


@pytest.mark.parametrize('B,slots,spare,units,layers,device_lengths', [(10, 4, 0, 24, 2, False), (40, 16, 5, 100, 1, True), (70, 32, 9, 600, 2, True),
                                                                       (32, 32, 3, 600, 1, True), (5, 1, 2, 8, 2, False)])
def test_pit_step_on_static_slots_vs_oracle(B, slots, spare, units, layers, device_lengths):
    """The row-slot layout whose length pattern is DEVICE data (``ops.sequence.StaticSlots``: fixed grid ``[steps, slots]``, index tables,
    row masks and frame counts as tensors; spare steps at the end idle in every slot) - what lets ONE captured step serve ragged
    batches: the whole step against the oracle as for ``SlotLayout``; ``device_lengths``: the feature kernel reads the sample counts
    from the device too and returns bit for bit what the host-side lengths give."""
    rng = np.random.RandomState(B + slots)
    n = 4400
    lens = sorted((int(x) for x in rng.randint(900, n + 1, B)), reverse=True)
    lens[0] = n
    # (B = 70: one ReLU tie of linear1 at a pre-activation of 1e-10 moves linear1.weight's gradient by 2.8e-4, profiles/r6_relu_tie.txt)
    _pit_case(B, 8000, lens=lens, seed=B, n=n, static_slots=(slots, spare), device_lengths=device_lengths, in_place=True, units=units,
              recurrent_layers=layers, K=2 + B % 2, grad_tol=5e-4 if B == 70 else 2e-4)


def test_row_slot_masks_equal_the_packed_sequence_path():
    """The same ragged batch through the PackedSequence path and through row slots: identical masks per example up to the
    rounding of differently tiled GEMMs (1e-6), and the model falls back to the PackedSequence path for equal lengths."""
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    torch.manual_seed(3)
    model = PermutationInvariantTrainingModel(F=257, recurrent_layers=2, units=40, K=2).to(DEV).eval()
    lens = [4000, 3600, 3300, 2800, 2100, 2000, 1500, 900, 700]
    s = _waveforms(len(lens), 2, 4000, lens, 5).to(DEV)
    feats = pt.ops.pit_features(s.sum(1), s, lens)
    with torch.no_grad():
        plain = model(feats)
        model.row_slots = 3
        slotted = model(feats)
        for a, b in zip(plain, slotted):
            assert a.shape == b.shape
            torch.testing.assert_close(a, b, atol=1e-6, rtol=0)
        equal = pt.ops.pit_features(s[:4].sum(1), s[:4], [4000] * 4)
        assert not model(equal).batch_first            # equal lengths: the time-major PackedSequence path, as before


def test_pit_step_config3_rows_and_steps_vs_oracle():
    """More than 32 sequences x 503 steps: the 32-row chains (two row tiles per workgroup) of the config-3 kernels; a few
    shorter examples make the batch ragged at the end (the hand-off bookkeeping of shrinking steps)."""
    n = 64000
    lens = [n] * 36 + [n - 128 * 7, n - 128 * 40, n - 128 * 41, n - 128 * 200]
    _pit_case(40, 16000, lens=lens, seed=3)


def _dc_case(B, K, n, lens, seed, padded_target=False, row_slots=None, double=True, **model_kw):
    import padertorch_amd as pt
    from padertorch_amd.contrib.tcl.dc import DeepClusteringModel
    from padertorch_amd.ops import lstm as _lstm
    from padertorch_amd.ops.sequence.pack_module import PaddedList
    from oracle import torch_ref
    torch.manual_seed(seed)
    model = DeepClusteringModel(**model_kw)
    ref = torch_ref.DCModelRef(**model_kw)
    ref.load_state_dict(model.state_dict())
    model.to(DEV).train()
    model.row_slots = row_slots
    s = _waveforms(B, K, n, lens, seed + 1).to(DEV)
    feats = pt.ops.pit_features(s.sum(1), s, lens)
    target = [torch.nn.functional.one_hot(x.argmax(1), K).permute(0, 2, 1).to(torch.float32) for x in feats['X_abs']]
    given = target
    if padded_target:        # what bench.py hands over: per-example views of ONE padded tensor
        X = feats['X_abs'].padded
        whole = torch.nn.functional.one_hot(X.argmax(2), K).permute(0, 1, 3, 2).to(torch.float32, memory_format=torch.contiguous_format)
        given = PaddedList(whole, feats['num_frames'], True, feats['Y_abs'].lengths_dev)
    batch = dict(Y_abs=feats['Y_abs'], target_mask=given, num_frames=feats['num_frames'])
    _lstm.CHECK_PERSISTENT_ERRORS = True
    try:
        emb = model(batch)
        loss = model.review(batch, emb)['losses']['dc_loss']
        loss.backward()
    finally:
        _lstm.CHECK_PERSISTENT_ERRORS = False
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    rb = dict(Y_abs=[t.detach().cpu() for t in feats['Y_abs']], target_mask=[t.cpu() for t in target])
    remb = ref(rb)
    rloss = ref.review(rb, remb)['losses']['dc_loss']
    rloss.backward()
    worst = max(float((m.detach().cpu() - r.detach()).abs().max()) for m, r in zip(emb, remb))
    assert worst < 1e-5, worst
    assert abs(float(loss) - float(rloss)) < 1e-4 * max(1., abs(float(rloss))), (float(loss), float(rloss))

    def run64(m64, to64):
        b64 = dict(Y_abs=[to64(t) for t in feats['Y_abs']], target_mask=[to64(t) for t in target])
        m64.review(b64, m64(b64))['losses']['dc_loss'].backward()
    _grad_check(model, ref, _double_oracle(ref, run64) if double else None)


@pytest.mark.parametrize('B,slots,transform', [(36, 16, 'log1p'), (9, 4, 'identity')])
def test_dc_step_on_row_slots_vs_oracle(B, slots, transform):
    """The deep-clustering model on row slots (2-3 sequences end to end per slot): embeddings, loss and every gradient against the
    oracle (one sequence per row), with the bench's PaddedList targets."""
    rng = np.random.RandomState(B)
    n = 4000
    lens = sorted((int(x) for x in rng.randint(900, n + 1, B)), reverse=True)
    lens[0] = n
    _dc_case(B, 3, n, lens, seed=B, padded_target=True, row_slots=slots, units=40, recurrent_layers=2, E=8, input_feature_transform=transform)


def test_dc_step_config5_shape_vs_oracle():
    n = 64000
    _dc_case(34, 3, n, [n] * 30 + [n - 128 * 3, n - 128 * 90, n - 128 * 91, n - 128 * 300], 5)


@pytest.mark.parametrize('seed', range(8))
def test_dc_step_random_small_configurations_vs_oracle(seed):
    """Random batch sizes, ragged / equal lengths, target masks as a plain list or as the bench's PaddedList over one tensor."""
    rng = np.random.RandomState(2000 + seed)
    B = int(rng.randint(1, 41))
    n = int(rng.randint(800, 4800))
    ragged = bool(rng.randint(0, 2))
    lens = sorted((int(x) for x in rng.randint(n // 3, n + 1, B)), reverse=True) if ragged else [n] * B
    lens[0] = n
    _dc_case(B, 3, n, lens, 40 + seed, padded_target=bool(seed % 2))


def test_pit_step_config1_batch4_vs_oracle():
    """BASELINE configs[0]: batch 4 x 4 s at 8 kHz (the reference's own CPU-runnable case): the 8-unit forward tiles of B <= 16."""
    _pit_case(4, 8000, seed=7)


def test_pit_step_config3_full_batch_vs_oracle():
    """BASELINE configs[2] at its FULL batch: 64 x 4 s at 16 kHz, equal lengths - exactly what `bench.py --config c3` times: 32-row
    chains, T = 503, the hand-off planes as GEMM operands (equal-length batch), the 256 x 320 tiles at M = 32192."""
    _pit_case(64, 16000, seed=9)


def test_dc_step_config5_full_batch_vs_oracle():
    """BASELINE configs[4] at its full batch (64 x 4 s at 16 kHz, K = 3, equal lengths): `bench.py --config c5` (incl. its PaddedList of
    target masks)."""
    n = 64000
    _dc_case(64, 3, n, [n] * 64, 15, padded_target=True)


def test_config4_four_micro_steps_one_optimizer_step_vs_oracle(tmp_path):
    """BASELINE configs[3], the work of ONE GPU (W = 1): batch 64 at 16 kHz, virtual_minibatch_size = 4 - four micro-steps whose
    gradients ACCUMULATE in the flat bucket (in place, side stream; 32-row chains), then one clip + Adam - through the Trainer,
    against the oracle's train_step arithmetic over the same four batches (trainer.py:357-393,512-532): the four losses (1e-4),
    the ACCUMULATED gradient of every parameter (2e-4 of its largest entry), the clipped norm, and the Adam update wherever it is
    well conditioned (first step: lr g / (|g| + eps) - entries with |g| >> eps move by lr sign(g)).  Signals of 0.5 s keep the
    oracle's four CPU passes at batch 64 affordable; the full 4 s step at batch 64 is test_pit_step_config3_full_batch_vs_oracle,
    and the data-parallel run adds only the all-reduce(SUM) of the bucket (tests/test_trainer.py, gloo)."""
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    from padertorch_amd.ops import lstm as _lstm
    from oracle import torch_ref
    B, n, micro = 64, 8000, 4
    lw = dict(pit_ips_loss=1., pit_mse_loss=0.)
    torch.manual_seed(21)
    model = PermutationInvariantTrainingModel()
    ref = torch_ref.PITModelRef()
    ref.load_state_dict(model.state_dict())
    before = {k: v.clone() for k, v in ref.state_dict().items()}
    trainer = pt.Trainer(model, tmp_path, pt.optimizer.Adam(gradient_clipping=1.), loss_weights=lw, virtual_minibatch_size=micro)
    trainer.to(torch.device(DEV))
    trainer._flat = trainer.optimizer.use_flat_grads()
    model.train()
    defer = _lstm.DEFER_WGRAD
    _lstm.DEFER_WGRAD = True
    _lstm.warm_side_stream(torch.device(DEV))
    losses, rbatches = [], []
    try:
        for m in range(micro):
            s = _waveforms(B, 2, n, [n] * B, 30 + m).to(DEV)
            feats = pt.ops.pit_features(s.sum(1), s, [n] * B)
            rbatches.append({k: [t.detach().cpu() for t in feats[k]] for k in ('Y_abs', 'X_abs', 'cos_phase_difference')})
            loss, _, _, _ = trainer.train_step(model, feats, torch.device(DEV))
            loss.backward()
            losses.append(float(loss))
            del feats, loss
        _lstm.sync_deferred()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
        summary = trainer.optimizer_step()
        torch.cuda.synchronize()
        _lstm.check_errors()
    finally:
        _lstm.DEFER_WGRAD = defer
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    opt = torch.optim.Adam(ref.parameters())
    rlosses = []
    for rb in rbatches:                                     # oracle.torch_ref.train_step, with a look at the gradients before the clip
        out = ref(rb)
        rl = torch_ref.review_to_loss(ref.review(rb, out), lw)
        rl.backward()
        rlosses.append(float(rl))
    for a, b in zip(losses, rlosses):
        assert abs(a - b) < 1e-4, (losses, rlosses)
    rgrads = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    for k, r in rgrads.items():
        scale = float(r.abs().max())
        assert float((grads[k].double() - r.double()).abs().max()) <= 2e-4 * scale + 1e-9, (k, scale)
    rnorm = float(torch.nn.utils.clip_grad_norm_(list(ref.parameters()), 1.))
    opt.step()
    gn = float(summary['scalars']['grad_norm'])
    assert abs(gn - rnorm) < 2e-4 * rnorm, (gn, rnorm)
    clip = min(1., 1. / (rnorm + 1e-6))
    for (k, v), vr in zip(model.state_dict().items(), ref.state_dict().values()):
        du, dr = v.cpu() - before[k], vr - before[k]
        g = rgrads[k].abs() * clip                         # |g| >> eps = 1e-8 and >> the gradient's own error: update = lr sign(g)
        firm = g > max(1e-6, 2e-3 * float(g.max()))
        if firm.any():
            assert float((du - dr)[firm].abs().max()) < 1e-5, (k, float((du - dr)[firm].abs().max()))
        assert float(du.abs().max()) <= 1.001e-3 + 1e-7            # no entry moves by more than lr
