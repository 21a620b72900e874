"""Trainer with the loop semantics of ``padertorch/train/trainer.py`` and an RCCL data-parallel path.

Mirrors (reference file:line)
  * constructor / attributes                      trainer.py:42-148
  * ``train`` main loop, virtual minibatch        trainer.py:205-465  (gradients are ACCUMULATED, :87)
  * ``step`` / ``train_step`` / ``validation_step`` trainer.py:534-565
  * ``_review_to_loss_and_summary``               trainer.py:567-638  (weighted loss sum, finiteness)
  * ``optimizer_step`` / ``clip_grad``            trainer.py:512-532, 740-780
  * ``validate``                                  trainer.py:467-510
  * ``state_dict`` / checkpoints                  trainer.py:789-886
  * ``test_run`` invariants                       train/runtime_tests.py:74-410 (section 3.1 of SURVEY.md)

Data parallelism.  The reference's multi-GPU branch (trainer.py:396-442) is single-process:
``replicate`` broadcasts all 93.9 MB of parameters to every GPU and ``ReduceAddCoalesced`` reduces all
gradients into GPU 0 on EVERY micro-step, driven by GIL-bound python threads.  Here one process
owns one MI355X (``torch.distributed``, backend "nccl" = RCCL over xGMI): of every group of W
consecutive examples rank j takes the j-th (as trainer.py:357-359,413-419 hands example j to
device j), gradients accumulate locally in one flat fp32 bucket over ``virtual_minibatch_size // W``
micro-steps and are exchanged ONCE per optimizer step by ``all_reduce(SUM)`` - a sum, not a mean, like the
reference's ``gather(...).sum()`` (:426-428) - issued per layer bucket (:class:`GradBuckets`) during the last
micro-step's backward pass, so that the 94 MB of xGMI traffic run under the remaining backward kernels.  Every rank then applies the identical
clip + Adam, so replicas stay bit-identical without any parameter broadcast after step 0.  A rank
without an example in the last partial group contributes zeros (:408).
"""
import itertools
import json
import os
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

from .optimizer import Optimizer
from .trigger import EndTrigger, IntervalTrigger

__all__ = ['Trainer', 'StopTraining']

ALLOWED_REVIEW_KEYS = {'loss', 'losses', 'scalars', 'histograms', 'audios', 'images', 'texts',
                       'figures', 'buffers', 'snapshots', 'timings'}


class StopTraining(Exception):
    pass


class _Summary:
    """In-memory stand-in for the SummaryHook: accumulates scalars, emits means (hooks.py:153-405)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.data = dict(scalars={}, histograms={}, audios={}, images={}, texts={}, figures={},
                         buffers={}, snapshots={})

    def update(self, review):
        for key, value in review.get('scalars', {}).items():      # floats, or staged 0-dim host tensors
            self.data['scalars'].setdefault(key, []).append(value)
        for kind in ('images', 'audios', 'texts', 'figures', 'histograms'):
            self.data[kind].update(review.get(kind, {}))


class GradBuckets:
    """Layer-aligned slices of the flat gradient buffer and their all-reduce schedule.

    A bucket = the parameters of one top-level module (one layer of an ``nn.LSTM``), contiguous in the flat
    buffer.  ``ready(params)`` is called when gradients are final (autograd's post-accumulate hooks, or the
    in-place accumulation of ``ops.lstm`` / ``ops.linear``); a bucket is reduced once all its parameters are
    ready AND every later bucket has been issued: all ranks - also one that ran no backward pass because the
    last group of examples was short - issue the collectives in the same order (last bucket first).

    Readiness is tracked per DISTINCT parameter.  The in-place accumulation announces every use of a module in the forward
    pass (``expect``, through ``ops.lstm.GRAD_USE_HOOK``) and reports every finished use in the backward pass, so a module
    that is applied twice in one forward pass (shared weights) is ready only after its LAST backward use: its bucket is
    never reduced while a later contribution is still to be added (autograd's post-accumulate hooks fire once per leaf and
    need no announcement).
    """

    def __init__(self, model, flat_grads):
        import re
        flat = flat_grads.flat
        names = {id(p): n for n, p in model.named_parameters()}
        self.flat = flat
        self.buckets = []          # [start, end, number of parameters]
        self.bucket_of = {}
        last_key, off = None, 0
        for p in flat_grads.params:
            name = names.get(id(p), '')
            m = re.search(r'_l(\d+)(_reverse)?$', name)
            key = (name.split('.')[0], m.group(1) if m else None)
            if key != last_key:
                self.buckets.append([off, off, 0])
                last_key = key
            b = self.buckets[-1]
            b[1] = off + p.numel()
            b[2] += 1
            self.bucket_of[id(p)] = len(self.buckets) - 1
            off += p.numel()
        self.reset()

    def reset(self):
        self.done = [set() for _ in self.buckets]    # ids of the parameters whose gradients are final
        self.uses = {}                           # id(parameter) -> announced in-place uses whose backward has not run yet
        self.next = len(self.buckets) - 1        # buckets are issued last to first
        self.works = []
        self.active = False                      # True during the last micro-step of an optimizer step

    def expect(self, params):
        """Forward pass of a module whose weight gradients will be accumulated in place: one more backward use to wait for."""
        if not self.active:
            return
        for p in params:
            if id(p) in self.bucket_of:
                self.uses[id(p)] = self.uses.get(id(p), 0) + 1

    def ready(self, params, stream=None):
        if not self.active:
            return
        for p in params:
            i = self.bucket_of.get(id(p))
            if i is None:
                continue
            left = self.uses.get(id(p), 0)
            if left > 1:                         # an earlier use of the same module is still to come in this backward pass
                self.uses[id(p)] = left - 1
                continue
            self.uses.pop(id(p), None)
            self.done[i].add(id(p))
        self._issue(stream, everything=False)

    def _issue(self, stream, everything):
        while self.next >= 0 and (everything or len(self.done[self.next]) >= self.buckets[self.next][2]):
            start, end, _ = self.buckets[self.next]
            seg = self.flat[start:end]
            if seg.is_cuda and stream is not None:
                # the collective has to wait for the accumulations on BOTH the main and the weight-gradient stream
                stream.wait_stream(torch.cuda.current_stream(seg.device))
                with torch.cuda.stream(stream):
                    self.works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))
            else:
                self.works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))
            self.next -= 1

    def finish(self):
        """Issue what is left (in order) and make the current stream / the host wait for every bucket."""
        self._issue(None, everything=True)
        for w in self.works:
            w.wait()
        self.reset()


class Trainer:
    def __init__(
            self,
            model,
            storage_dir,
            optimizer,
            loss_weights=None,
            summary_trigger=(1, 'epoch'),
            checkpoint_trigger=(1, 'epoch'),
            stop_trigger=(1, 'epoch'),
            virtual_minibatch_size=1,
            overlap_wgrad=True,
            deferred_checks=False,
            overlap_allreduce=True,
            graph_steps=False,
    ):
        if not isinstance(model, torch.nn.Module):
            raise TypeError('Expect that the model is a subclass from padertorch.Module.\n'
                            f'Got: type: {type(model)}\n{model}')
        self.model = model
        assert isinstance(optimizer, Optimizer), optimizer
        optimizer.set_parameters(model.parameters())
        self.optimizer = optimizer
        self.device = None
        self.storage_dir = Path(storage_dir).expanduser().resolve()
        self.iteration = -1
        self.epoch = -1
        self.loss_weights = loss_weights
        self.virtual_minibatch_size = virtual_minibatch_size
        #: LSTM weight gradients accumulate in place; on a side stream, next to the next layer's recurrence,
        #: for the shapes whose GEMMs are pinned to kernels without inter-workgroup waits (ops.lstm.DEFER_WGRAD)
        self.overlap_wgrad = overlap_wgrad
        #: False: the loss and the gradient norm cross to the host in the step they belong to (two device
        #: syncs per step, as in the reference).  True: they are staged into pinned memory and inspected one
        #: optimizer step later; the optimizer update itself is gated ON THE DEVICE by their finiteness
        #: (fused optimizers' ``found_inf``), so a non-finite step still leaves the parameters untouched and
        #: raises the reference's RuntimeError -- one iteration late.  The host then runs ahead of the GPU.
        #: 'step': the same staging and device gating, inspected at the END of the same optimizer step (one host sync per step, behind
        #: the update's launch): the error surfaces in the iteration it belongs to, as in the reference.
        self.deferred_checks = deferred_checks
        #: data parallel: the gradient bucket of a layer is all-reduced as soon as that layer's gradients of the LAST
        #: micro-step of the optimizer step are complete, under the rest of the backward pass (False: one all-reduce
        #: of the whole flat buffer in optimizer_step)
        self.overlap_allreduce = overlap_allreduce
        #: True (one process, GPU): an optimizer step whose examples have been seen before - same structure, shapes, lengths - runs as ONE
        #: captured hipGraph (``train.graphed.GraphedStep``): no python between its ~120 launches, the loss / gradient-norm / watchdog
        #: checks at the end of the SAME step at the cost of one synchronisation (whatever ``deferred_checks`` says: a replayed step
        #: raises in its own iteration); the first step of every new shape runs eagerly, the second one captures.  Ragged data with
        #: ever-new length patterns stays eager.
        self.graph_steps = graph_steps
        #: optional ``example -> model input`` captured inside the graph in front of ``train_step`` (a feature front-end on the device)
        self.graph_prepare = None
        self._graphs = {}            # signature -> GraphedStep (at most two: a graph owns the memory of a whole step)
        self._graph_seen = {}
        self._graph_refused = set()  # signatures whose capture raised: eager from then on
        self._graph_misses = 0
        self.graph_capacity = 2
        self.graph_eviction_cooldown = 32        # optimizer steps between two evictions
        self._graph_evicted_at = -(1 << 30)
        #: this Trainer's switches / hooks of the in-place weight-gradient path, attached to every module of its model
        #: (ops.context): nothing process-global is set or reset around train()
        from ..ops import context as _context
        self.op_context = _context.attach(self.model, _context.OpContext())
        self._buckets = None
        self._pending = []           # [(what, event, host tensor, context, optimizer step)]
        self._stage_queue = []       # staged scalars whose copies are not enqueued yet (_flush_stage)
        self._graph_stage = None     # train.graphed: the static pinned words of a step that is being captured
        self._opt_step = 0
        self._loss_acc = None        # device scalar: sum of the losses of the current optimizer step (non-finite -> skip)
        #: how the ranks exchange an optimizer step's results (data parallel).  None: the layer buckets / one all-reduce of the flat
        #: bucket plus a MAX all-reduce of the update gate (the eager loop).  'flat+words' (``graph_steps`` with a process group): exactly
        #: TWO collectives per optimizer step on every rank whatever the rank does in between - all_reduce(SUM) of the flat bucket, then
        #: all_reduce(SUM) of two words [sum of the step's losses, recurrence watchdog count] - so that a rank that replays a captured
        #: step (``train.graphed``: graph A = forward + backward, the exchange, graph B = norm + clip + Adam) and a rank that runs the
        #: same step eagerly (first sighting of a shape) stay in lockstep; the update is gated on the device by the SUMMED loss word.
        self.dp_protocol = None
        #: how a captured step (``graph_steps``) exchanges under a process group: 'split' (default) = two graphs around the 'flat+words'
        #: exchange, nothing of RCCL captured; 'captured' (opt-in, validated with a one-rank group only) = one graph that contains the layer
        #: buckets' all-reduces and the update gate's - the eager loop's overlap inside the replay
        self.graph_exchange = 'split'
        self._exchanged = None       # the words of an exchange that has run for the current optimizer step
        self.summary_trigger = IntervalTrigger.new(summary_trigger)
        self.checkpoint_trigger = IntervalTrigger.new(checkpoint_trigger)
        self.stop_trigger = EndTrigger.new(stop_trigger)
        self.validation_iterator = None
        self.validation_metric = 'loss'
        self.train_summary = _Summary()
        self.summaries = []          # list of (iteration, prefix, dict of mean scalars)
        self.timer = {}
        self._flat = None

    # ------------------------------------------------------------------ distributed helpers
    @property
    def world_size(self):
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    @staticmethod
    def _dp_active():
        return dist.is_available() and dist.is_initialized()

    @staticmethod
    def backward(loss):
        """``loss.backward(retain_graph=False)`` (``trainer.py:375``) starting from a cached ``1.`` instead of a ``ones_like`` fill
        launch per micro-step (``ops.scalars``)."""
        from ..ops import scalars as _scalars
        one = _scalars.unit_grad(loss) if loss.is_cuda else None
        if one is None:
            loss.backward(retain_graph=False)
        else:
            torch.autograd.backward(loss, one, retain_graph=False)

    def _time(self, key, t0):
        self.timer[key] = self.timer.get(key, 0.) + time.perf_counter() - t0

    # ------------------------------------------------------------------ public API
    def register_validation_hook(self, validation_iterator, metric='loss', maximize=False):
        """Validate at every checkpoint trigger and track the best checkpoint (trainer.py:699-738)."""
        self.validation_iterator = validation_iterator
        self.validation_metric = metric
        self.validation_maximize = maximize
        self._best = None          # a resume restores it from the checkpoint (load_state_dict runs after this)

    def to(self, device):
        self.model.to(device)
        self.optimizer.to(device)
        self.device = device
        return self

    def train(self, train_dataset, *, resume=False, device=None):
        """Same contract as ``pt.Trainer.train`` (trainer.py:205-465); ``device`` is ONE device per
        process (``int`` / ``str`` / ``torch.device``); multi-GPU = launch one process per GPU."""
        if isinstance(device, (list, tuple)):
            assert len(device) == 1, (
                'padertorch_amd runs one process per GPU (torchrun); a device list is the '
                "reference's single-process mode", device)
            device = device[0]
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else 'cpu'
        if resume:
            assert self.checkpoint_dir.exists(), self.checkpoint_dir
            self.load_checkpoint()
        else:
            assert not self.checkpoint_dir.exists() or not any(self.checkpoint_dir.iterdir()), \
                f'A checkpoint directory already exists ({self.checkpoint_dir}); use resume=True'
            self.iteration, self.epoch = 0, 0
            for trigger in (self.summary_trigger, self.checkpoint_trigger, self.stop_trigger):
                trigger.set_last(-1, -1)
        self.model.train()
        self.to(device)
        W = self.world_size
        assert self.virtual_minibatch_size % W == 0, (self.virtual_minibatch_size, W)
        self._flat = self.optimizer.use_flat_grads()
        if W > 1:
            self._broadcast_parameters()
        self.optimizer.zero_grad()
        from ..ops import lstm as _lstm
        # this model's own switches and hooks (ops.context): another Trainer in the same process - a second model trained
        # alternately, an EMA / validation copy, a model per host thread - has its own
        oc = self.op_context
        defer_before = oc.defer_wgrad
        oc.defer_wgrad = bool(self.overlap_wgrad) and self._flat.flat.is_cuda
        if oc.defer_wgrad:
            _lstm.warm_side_stream(self._flat.flat.device)
        # graph_steps with a process group: the 'flat+words' protocol (see dp_protocol) for EVERY optimizer step of this run - replayed or
        # eager -, the checks at the end of the step they belong to (what a replayed step does anyway); no layer buckets
        dp_graph = bool(self.graph_steps and self._dp_active() and self._flat.flat.is_cuda)
        if dp_graph and not (getattr(self.optimizer, '_native_ok', None) is not None and self.optimizer._native_ok()):
            import warnings
            warnings.warn('graph_steps with a process group needs the native Adam step on a GPU bucket; running the eager data-parallel loop')
            dp_graph = False
        checks_before, protocol_before = self.deferred_checks, self.dp_protocol
        if dp_graph and self.graph_exchange == 'captured':
            self.deferred_checks = 'step'           # (the eager loop's collectives, captured: train.graphed.GraphedStep.captured_exchange)
        elif dp_graph:
            self.dp_protocol, self.deferred_checks = 'flat+words', 'step'
        if self.dp_protocol == 'flat+words':
            hooks, self._buckets = [], None         # ONE all-reduce of the flat bucket per optimizer step: no layer buckets
        else:
            hooks = self.enable_bucketed_allreduce()

        try:
            train_iterable = None
            while True:
                new_epoch = False
                if train_iterable is None:
                    new_epoch = True
                    self._pre_step()            # hooks run between the epochs (trainer.py:348-353)
                    train_iterable = iter(train_dataset)
                optimize = True
                if self.graph_steps and self._flat.flat.is_cuda and (dp_graph or not self._dp_active()):
                    t0 = time.perf_counter()
                    whole = list(itertools.islice(train_iterable, self.virtual_minibatch_size))
                    self._time('time_per_data_loading', t0)
                    if len(whole) == self.virtual_minibatch_size:
                        if new_epoch:
                            new_epoch = False
                        else:
                            self._pre_step()
                        t0 = time.perf_counter()
                        # (data parallel: of every group of W consecutive examples rank j takes the j-th, as the plain loop below does)
                        self._graph_or_eager_step(whole[self.rank::W], device)
                        self._time('time_per_optimize', t0)
                        self.iteration += 1
                        continue
                    train_iterable = iter(whole)        # the epoch ends inside this group: the plain loop takes what is left
                for minibatch_index in range(self.virtual_minibatch_size // W):
                    if self._buckets is not None:
                        self._buckets.active = minibatch_index + 1 == self.virtual_minibatch_size // W
                    t0 = time.perf_counter()
                    group = list(itertools.islice(train_iterable, W))
                    self._time('time_per_data_loading', t0)
                    if len(group) == 0:
                        train_iterable = None
                        self.epoch += 1
                        if minibatch_index == 0:
                            optimize = False
                        break
                    if new_epoch:
                        new_epoch = False
                    elif minibatch_index == 0:
                        self._pre_step()
                    if self.rank < len(group):
                        self._collective_loss_check = W > 1 and not self.deferred_checks
                        loss, example, model_output, review = self.train_step(
                            self.model, group[self.rank], device)
                        self.train_summary.update(review)
                        del example, model_output, review
                        t0 = time.perf_counter()
                        self.backward(loss)
                        self._time('time_per_backward', t0)
                        del loss
                    elif W > 1 and not self.deferred_checks:
                        self._all_ranks_finite(True)      # takes part in the other ranks' loss check
                    # else: idle rank of a partial last group: contributes zero gradient (:408)
                if optimize:
                    t0 = time.perf_counter()
                    summary = self.optimizer_step()
                    self.train_summary.update(summary)
                    self._time('time_per_optimize', t0)
                    self.iteration += 1
        except StopTraining:
            pass
        finally:
            for h in hooks:
                h.remove()
            oc.grad_ready_hook = None
            oc.grad_use_hook = None
            self._buckets = None
            self.deferred_checks, self.dp_protocol, self._exchanged = checks_before, protocol_before, None
            _lstm.sync_deferred()
            oc.defer_wgrad = defer_before
            opt = self.optimizer.optimizer
            if getattr(opt, 'found_inf', None) is not None:
                opt.found_inf = None
            try:
                self._check_pending(flush=True)
            finally:
                self._graphs, self._graph_seen, self._graph_refused = {}, {}, set()
                self._graph_evicted_at = -(1 << 30)
                self._close()

    def _graph_or_eager_step(self, group, device):
        """One optimizer step on the ``virtual_minibatch_size`` examples ``group`` (``graph_steps``): replayed from the graph of their
        signature, captured at the second sighting of a signature, eager before that and for examples no graph can take."""
        from .graphed import GraphedStep, signature
        examples = [self.model.example_to_device(e, device) for e in group]
        sig = signature(examples)
        if sig in self._graph_refused:
            sig = None
        graphed = self._graphs.get(sig) if sig is not None else None
        if graphed is not None:
            self._graphs[sig] = self._graphs.pop(sig)       # (most recently used last)
        room = len(self._graphs) < self.graph_capacity
        if not room and self._opt_step - self._graph_evicted_at >= self.graph_eviction_cooldown:
            # a graph keeps a whole step's memory: the least recently used one goes - at most once per cool-down: with more recurring
            # shapes than graphs every step would otherwise evict one graph and capture another (a capture costs several eager
            # steps); the shapes without a graph run eagerly meanwhile
            room = 'evict'
        if graphed is None and sig is not None and self._graph_seen.get(sig, 0) >= 1 and room:
            if room == 'evict':
                self._graphs.pop(next(iter(self._graphs)))
                self._graph_evicted_at = self._opt_step
            try:
                graphed = self._graphs[sig] = GraphedStep(self, examples, prepare=self.graph_prepare, warmup=0, clone_inputs=True)
            except Exception as e:      # noqa: a step that cannot be captured (an optimizer that synchronises, a host read in the model)
                graphed = None
                self._graph_refused.add(sig)
                self._abandon_capture(device)
                import warnings
                warnings.warn(f'graph_steps: this step cannot be captured into a hipGraph ({type(e).__name__}: {e}); '
                              'steps of this shape run eagerly')
        if graphed is not None:
            self._graph_misses = 0
            return graphed(examples)
        if sig is not None:
            if len(self._graph_seen) > 64:
                self._graph_seen.clear()
            self._graph_seen[sig] = self._graph_seen.get(sig, 0) + 1
        self._graph_misses += 1
        if self._graph_misses == 50 and not self._graphs:
            import warnings
            warnings.warn('graph_steps: 50 optimizer steps without two examples of one signature (shapes, lengths, python numbers of the '
                          'examples): every step runs eagerly.  Ragged batches need fixed shapes to share a graph: model.row_slots / '
                          'padded buckets.')
        for example in examples:
            batch = self.graph_prepare(example) if self.graph_prepare is not None else example
            loss, _, _, review = self.train_step(self.model, batch, device)
            self.train_summary.update(review)
            self.backward(loss)
            del loss, review, batch
        summary = self.optimizer_step()
        self.train_summary.update(summary)
        return summary

    def _abandon_capture(self, device):
        """After a capture that raised: nothing of it has run, but the host-side queues of the step path still name its tensors and
        events (weight-gradient closures waiting for their enqueue point, staged scalars, the device flag of the update gate)."""
        from ..ops import capture as _capture, lstm as _lstm
        del _lstm._PENDING_WGRAD[:]
        self._stage_queue = []
        self._graph_stage = None
        self._loss_acc = None
        if self._buckets is not None:
            self._buckets.reset()
        opt = self.optimizer
        if getattr(opt, 'skip_if_not_finite', None) is not None:
            opt.skip_if_not_finite = None
        if hasattr(opt, '_norm'):
            opt._norm = None
        if hasattr(opt, 'hyper_from_device'):
            opt.hyper_from_device = False
        _capture.ACTIVE = False
        _capture.reset_step_caches()
        torch.cuda.synchronize(device)
        # (gradients: the capture executed nothing - the bucket is what the last zero_grad left)

    # ------------------------------------------------------------------ hooks (fixed set)
    def _pre_step(self):
        """Summary(50) > Validation(20) > Checkpoint(11) > Stop(10) priority order (hooks.py:43-62)."""
        it, ep = self.iteration, self.epoch
        if self.summary_trigger(it, ep) and it > 0:
            self._dump_summary('training')
        if self.checkpoint_trigger(it, ep):
            if self.validation_iterator is not None:
                self._run_validation()
            if self.rank == 0:
                self.save_checkpoint()
        if self.stop_trigger(it, ep):
            raise StopTraining

    def _close(self):
        self._dump_summary('training')
        if self.rank == 0 and self.iteration >= 0:
            if self.validation_iterator is not None and not self.default_checkpoint_path().exists():
                self._run_validation()
            if not self.default_checkpoint_path().exists():
                self.save_checkpoint()

    def _dump_summary(self, prefix, summary=None):
        summary = summary or self.train_summary
        data = summary.data
        if not data['scalars']:
            summary.reset()
            return None
        self._check_pending(flush=True)        # staged scalars (deferred_checks) are final after this
        data['scalars'] = {k: [float(x.item() if torch.is_tensor(x) else x) for x in v]
                           for k, v in data['scalars'].items()}
        data = self.model.modify_summary(data) if hasattr(self.model, 'modify_summary') else data
        scalars = {k: float(np.mean(v)) for k, v in data['scalars'].items()}
        self.summaries.append((self.iteration, prefix, scalars))
        if self.rank == 0:
            self.storage_dir.mkdir(parents=True, exist_ok=True)
            with open(self.storage_dir / 'summary.jsonl', 'a') as f:
                f.write(json.dumps(dict(iteration=self.iteration, epoch=self.epoch, prefix=prefix,
                                        scalars=scalars)) + '\n')
        summary.reset()
        return scalars

    def _run_validation(self):
        val = _Summary()
        for _, _, review in self.validate(self.validation_iterator):
            val.update(review)
        scalars = self._dump_summary('validation', val) or {}
        metric = scalars.get(self.validation_metric)
        if metric is not None and self.rank == 0:
            better = self._best is None or (metric > self._best[0] if self.validation_maximize
                                            else metric < self._best[0])
            if better:
                self._best = (metric, self.iteration)
        return scalars

    # ------------------------------------------------------------------ step path
    def validate(self, validation_iterator):
        """trainer.py:467-510: eval mode, no_grad, yields (example, model_out, review)."""
        train_end_time = self.model.training
        self.model.eval()
        try:
            with torch.no_grad():
                for example in validation_iterator:
                    yield self.validation_step(self.model, example, self.device)
            from ..ops import lstm as _lstm
            _lstm.check_errors()        # the watchdog words of the persistent LSTM kernels (nothing else reads them here)
        finally:
            self.model.train(train_end_time)

    def optimizer_step(self):
        """clip (global norm) -> lr summary -> optimizer.step -> zero_grad (trainer.py:512-532).
        With W > 1 the flat gradient bucket is summed over all ranks first (one collective)."""
        from ..ops import lstm as _lstm
        if self._exchanged is None:
            _lstm.sync_deferred()      # side-stream weight-gradient accumulations (ops.lstm.DEFER_WGRAD)
        # (else: the second graph of a captured data-parallel step - the first one has joined the weight-gradient queue and the exchange
        #  has run behind it; a wait for a queue that holds nothing of THIS capture would be no edge of the graph)
        if self._dp_active():
            t0 = time.perf_counter()
            if self.dp_protocol == 'flat+words':
                if self._exchanged is None:        # (a captured step has run its exchange between its two graphs already)
                    self._exchanged = self._exchange()
            elif self._buckets is not None:
                self._buckets.finish()             # buckets not yet issued + wait for all of them
            else:
                dist.all_reduce(self._flat.flat, op=dist.ReduceOp.SUM)
            self._time('time_per_all_reduce', t0)
        summary = self.clip_grad({})
        for i, param_group in enumerate(self.optimizer.optimizer.param_groups):
            summary['scalars'][f'lr/param_group_{i}'] = param_group['lr']
        if hasattr(self.optimizer, 'step_and_zero_grad'):
            self.optimizer.step_and_zero_grad()        # Adam on a GPU bucket: clip + update + zeroing in one kernel
        else:
            self.optimizer.step()
            self.optimizer.zero_grad()
        self._flush_stage()            # this step's staged scalars (deferred_checks): copied behind the update
        self._opt_step += 1
        if self.deferred_checks == 'step':
            # ONE host sync per optimizer step, behind everything the step has enqueued: the loss / gradient-norm / watchdog
            # checks raise in the iteration they belong to, like the reference's (whose two syncs sit in the MIDDLE of the step:
            # after the loss and after the norm); the update itself was gated on the device, so the parameters are what the
            # reference leaves behind when it raises before optimizer.step()
            self._check_pending(flush=True)
        return summary

    def clip_grad(self, summary: dict):
        """trainer.py:740-780 incl. the non-finite check (one host sync per optimizer step)."""
        summary.setdefault('scalars', {})
        summary.setdefault('histograms', {})
        from ..ops import lstm as _lstm
        grad_norm = self.optimizer.clip_grad()
        exchanged, self._exchanged = self._exchanged, None
        if self._deferred(grad_norm):
            self._check_pending()                          # the PREVIOUS optimizer step's loss / norm / watchdog
            loss_acc, self._loss_acc = self._loss_acc, None
            opt = self.optimizer.optimizer
            native = getattr(self.optimizer, '_native_ok', None)
            if exchanged is not None:
                # 'flat+words': every rank holds the same summed gradients, the same sum of all ranks' losses and the same watchdog
                # count - the same gate and the same checks everywhere without another collective
                assert native is not None and native(), "dp_protocol 'flat+words' needs the native Adam step (csrc/optim.hip)"
                self.optimizer.skip_if_not_finite = exchanged[0]
                opt.found_inf = None
                host = self._stage('grad_norm', [grad_norm.detach().reshape(1), exchanged[1:2].to(torch.int32), exchanged[0:1]], summary)
                summary['scalars']['grad_norm'] = host[0][0]
                summary['histograms']['grad_norm_'] = host[0]
                return summary
            if self.world_size == 1 and native is not None and native():
                # the update kernel itself skips on a non-finite gradient norm or loss sum (csrc/optim.hip): no flag kernels
                self.optimizer.skip_if_not_finite = loss_acc
                opt.found_inf = None
            else:
                bad = ~torch.isfinite(grad_norm)
                if loss_acc is not None:
                    bad = bad | ~torch.isfinite(loss_acc)
                found = bad.to(torch.float32)
                timeouts = _lstm.error_count(grad_norm.device).reshape(1)
                if self._dp_active():
                    # a rank-local non-finite loss skips the update everywhere, and a timed-out recurrence launch on ONE rank is
                    # seen by ALL of them one step later (same branch everywhere: nobody is left waiting in a collective)
                    both = torch.cat([found.reshape(1), timeouts.to(torch.float32)])
                    dist.all_reduce(both, op=dist.ReduceOp.MAX)
                    found, timeouts = both[0], both[1:2].to(torch.int32)
                opt.found_inf, opt.grad_scale = found, None    # gates optimizer.step on the device
                host = self._stage('grad_norm', [grad_norm.detach().reshape(1), timeouts], summary)
                summary['scalars']['grad_norm'] = host[0][0]
                summary['histograms']['grad_norm_'] = host[0]
                return summary
            host = self._stage('grad_norm', [grad_norm.detach().reshape(1), _lstm.error_count(grad_norm.device).reshape(1)],
                               summary)
            summary['scalars']['grad_norm'] = host[0][0]
            summary['histograms']['grad_norm_'] = host[0]
            return summary
        grad_norm = float(grad_norm)                       # host sync
        if exchanged is not None:
            # 'flat+words' without device-gated checks (a CPU bucket, an optimizer without a fused step): the summed words on the host
            loss_sum, count = (float(v) for v in exchanged.tolist())
            if self._flat.flat.is_cuda and _lstm.errors_since_last_report(self._flat.flat.device, int(count)):
                _lstm.raise_timeout(self._flat.flat.device)
            if not np.isfinite(loss_sum):
                raise RuntimeError('The loss of another rank is not finite (see its error state).')
        elif self._dp_active() and self._flat is not None and self._flat.flat.is_cuda:
            # persistent-kernel watchdog: every rank learns about a timeout on ANY rank before the next collective
            cnt = _lstm.error_count(self._flat.flat.device).reshape(1).clone()
            dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
            if _lstm.errors_since_last_report(self._flat.flat.device, int(cnt)):
                _lstm.raise_timeout(self._flat.flat.device)
        else:
            _lstm.check_errors()                           # persistent-kernel watchdog words
        if not np.isfinite(grad_norm):
            path = self.log_error_state({'state_dict': self.state_dict(), 'optimizer_summary': summary})
            raise RuntimeError(f'The grad_norm ({grad_norm}) is not finite.\n'
                               f'See error states (model, example, model_out and review) in {path}.')
        summary['scalars']['grad_norm'] = grad_norm
        summary['histograms']['grad_norm_'] = torch.Tensor([grad_norm])
        return summary

    def _local_words(self):
        """[sum of this rank's losses of the step, watchdog count] as a device fp32 [2] (consumes the running loss sum)."""
        from ..ops import lstm as _lstm
        dev = self._flat.flat.device
        loss_acc, self._loss_acc = self._loss_acc, None
        loss = loss_acc.detach().reshape(1).to(torch.float32) if loss_acc is not None else torch.zeros(1, device=dev)   # (an idle rank)
        count = _lstm.error_count(dev).reshape(1).to(torch.float32) if dev.type == 'cuda' else torch.zeros(1, device=dev)
        return torch.cat([loss, count])

    def _exchange(self, words=None):
        """The 'flat+words' protocol's two collectives; returns the summed words."""
        if words is None:
            words = self._local_words()
        dist.all_reduce(self._flat.flat, op=dist.ReduceOp.SUM)
        dist.all_reduce(words, op=dist.ReduceOp.SUM)
        return words

    def train_step(self, model, example, device):
        return self.step(model, example, device, 'train')

    def validation_step(self, model, example, device):
        # [1:] -> ignore the loss. Is already in scalars.
        return self.step(model, example, device, 'validate')[1:]

    def step(self, model, example, device, mode='train'):
        """to_device -> forward -> review -> loss (trainer.py:541-565); dumps an error state and
        re-raises on any exception."""
        try:
            t0 = time.perf_counter()
            example = model.example_to_device(example, device)
            self._time('time_per_to_device', t0)
            t0 = time.perf_counter()
            model_out = model(example)
            self._time('time_per_forward', t0)
            t0 = time.perf_counter()
            review = model.review(example, model_out)
            loss, summary = self._review_to_loss_and_summary(review)
            self._time('time_per_review', t0)
            return loss, example, model_out, summary
        except Exception:
            from ..ops import capture as _capture
            if _capture.ACTIVE:
                raise       # inside a stream capture a state dump (device-to-host copies) would fail itself and mask this error
            data = {'state_dict': self.state_dict(), 'example': example}
            if 'model_out' in locals():
                data['model_out'] = model_out
            if 'review' in locals():
                data['review'] = review
            path = self.log_error_state(data)
            print(f'Wrote\n{path}\nfor debugging.')
            raise

    def _review_to_loss_and_summary(self, review):
        """trainer.py:567-638.  ``loss = sum_k loss_weights[k] * losses[k]`` over non-zero weights;
        every loss is logged.  All scalars of the step cross to the host in ONE transfer (the
        reference does one ``.item()`` per loss, i.e. one device sync each)."""
        assert set(review) <= ALLOWED_REVIEW_KEYS, set(review) - ALLOWED_REVIEW_KEYS
        review.setdefault('scalars', {})
        if 'losses' in review:
            assert 'loss' not in review, review
            losses = review['losses']
            loss_weights = self.loss_weights
            if len(losses) != 1:
                if loss_weights is None:
                    raise Exception('You can not have multiple losses without specifying '
                                    f'loss_weights. losses: {losses}')
                if set(loss_weights.keys()) != set(losses.keys()):
                    raise Exception('You can not have multiple losses without specifying '
                                    f'a loss_weight for each loss.\nlosses: {losses}\n'
                                    f'loss_weights: {loss_weights}')
            loss = 0.
            for key, value in losses.items():
                weight = loss_weights[key] if loss_weights is not None else 1.
                if weight != 0:
                    # 0. + 1 * value == value bit for bit: no kernels (forward or backward) for the trivial factors
                    if weight != 1 and self._graph_stage is not None:
                        # a captured step (train.graphed): the factor is a DEVICE word the replaying loop rewrites when
                        # trainer.loss_weights changes (hooks.py:957-966) - the same fp32 product as `weight * value`
                        term = self._graph_stage.loss_weight(key, weight) * value
                    else:
                        term = value if weight == 1 else weight * value
                    loss = term if (isinstance(loss, float) and loss == 0.) else loss + term
                review['scalars'][f'{key}_loss_weight'] = weight
            keys = list(losses)
            vals, index = self._loss_values(losses, keys, loss)
            if self._deferred(loss):
                host = self._stage('loss', vals, review)
                for i, k in zip(index, keys):
                    review['scalars'][k] = host[i]
                review['scalars']['loss'] = host[index[-1]]
                del review['losses']
                assert loss.dim() == 0, loss
                return loss, review
            host = vals.tolist()
            host = [host[i] for i in index]
            for k, v in zip(keys, host[:-1]):
                review['scalars'][k] = v
            loss_value = host[-1]
            del review['losses']
        else:
            assert 'loss' in review, review
            loss = review.pop('loss')
            if self._deferred(loss):
                review['scalars']['loss'] = self._stage('loss', loss.detach().reshape(1), review)[0]
                assert loss.dim() == 0, loss
                return loss, review
            loss_value = loss.item()
        review['scalars']['loss'] = loss_value
        assert loss.dim() == 0, loss
        finite = bool(np.isfinite(loss_value))
        if getattr(self, '_collective_loss_check', False) and self.model.training:
            # data parallel: every rank learns about a non-finite loss on ANY rank before the next collective, so that
            # all of them raise instead of one raising and the others blocking in the gradient all-reduce
            everyone = self._all_ranks_finite(finite)
            if finite and not everyone:
                raise RuntimeError('The loss of another rank is not finite (see its error state).')
        if not finite:
            path = self.log_error_state({'state_dict': self.state_dict(), 'review': review})
            raise RuntimeError(f'The loss ({loss_value}) is not finite.\n'
                               f'See error states (model, example, model_out and review) in {path}.')
        return loss, review

    @staticmethod
    def _loss_values(losses, keys, loss):
        """One tensor with every value the summary wants (each loss and the weighted sum) plus their positions in it - without a
        ``cat`` launch when the values already lie in one tensor: ``ops.scalars.pick``s of one loss vector whose weighted sum is one
        of them (weights 0 / 1, ``pit/train.py:68-71``), or a single loss."""
        from ..ops import scalars as _scalars
        tensors = [losses[k] for k in keys] + [loss]
        if all(t is tensors[0] for t in tensors):
            return tensors[0].detach().reshape(1), [0] * len(tensors)
        picks = [_scalars.picked_from(t) for t in tensors]
        # (the weighted sum stays the LAST staged value: what the update is gated on and the host checks read as [-1])
        if all(p is not None and p[0] is picks[0][0] for p in picks) and picks[-1][1] == picks[0][0].numel() - 1:
            return picks[0][0].detach(), [p[1] for p in picks]
        return torch.stack([t.detach().reshape(()) for t in tensors]), list(range(len(tensors)))

    # ------------------------------------------------------------------ deferred host checks
    def _deferred(self, t):
        """deferred_checks needs a CUDA step and an optimizer whose update can be gated on the device."""
        opt = getattr(self.optimizer, 'optimizer', None)
        return bool(self.deferred_checks and torch.is_tensor(t) and t.is_cuda and self.model.training
                    and opt is not None and opt.defaults.get('fused'))

    def _stage(self, what, vals, context):
        """Asynchronous device -> pinned host copy of a few scalars (a tensor, or a list of tensors that each keep their
        dtype); ``_check_pending`` inspects them later.  The copies themselves are enqueued by ``_flush_stage`` behind the
        optimizer kernel: in front of it (loss values between the loss and the backward pass, the gradient norm between the norm
        and the update) every copy is a blit launch of ~5 us on the step's critical path.
        The returned host tensors are valid only behind ``_check_pending(flush=True)`` (or the next ``optimizer_step``): until
        their copy has run they read NaN (floating point) / the type's minimum (integers), never stale memory, and the
        queue is bounded - code that calls ``train_step`` without ever reaching ``optimizer_step`` has its oldest entries flushed."""
        def blank(shape, dtype):
            if self._graph_stage is not None:       # a captured step (train.graphed): fixed addresses, no allocation inside the capture
                return self._graph_stage.blank(tuple(shape), dtype)
            h = torch.empty(shape, dtype=dtype, pin_memory=True)
            return h.fill_(float('nan')) if dtype.is_floating_point else h.fill_(torch.iinfo(dtype).min)
        if len(self._stage_queue) >= 64:
            self._flush_stage()
        if isinstance(vals, (list, tuple)):
            vals = [v.detach() for v in vals]
            host = [blank(v.shape, v.dtype) for v in vals]
        else:
            vals = vals.detach().to(torch.float32)
            if what == 'loss':      # a non-finite loss makes the sum non-finite: what the optimizer update is gated on
                self._loss_acc = vals[-1] if self._loss_acc is None else self._loss_acc + vals[-1]
            host = blank(vals.shape, torch.float32)
        self._stage_queue.append((what, vals, host, context, self._opt_step))
        if self._graph_stage is not None:
            self._graph_stage.jobs.append((what, host, context, vals))
        return host

    def _flush_stage(self):
        """Enqueue the staged copies (one event behind all of them)."""
        jobs, self._stage_queue = self._stage_queue, []
        if not jobs:
            return
        for _, vals, host, _, _ in jobs:
            for h, v in (zip(host, vals) if isinstance(host, list) else ((host, vals),)):
                h.copy_(v, non_blocking=True)
        if self._graph_stage is not None:
            return                          # (captured copy nodes: whoever replays the graph synchronises and inspects)
        event = torch.cuda.Event()
        event.record()
        for what, _, host, context, opt_step in jobs:
            self._pending.append((what, event, host, context, opt_step))

    def _check_pending(self, flush=False):
        """Raise the reference's errors for the staged values of EARLIER optimizer steps (all with flush)."""
        if flush:
            self._flush_stage()
        todo = [p for p in self._pending if flush or p[4] < self._opt_step]
        self._pending = [p for p in self._pending if not (flush or p[4] < self._opt_step)]
        for what, event, host, context, _ in todo:
            event.synchronize()
            if what == 'loss' and not np.isfinite(float(host[-1])):
                path = self.log_error_state({'state_dict': self.state_dict(), 'review': context})
                raise RuntimeError(f'The loss ({float(host[-1])}) is not finite.\n'
                                   f'See error states (model, example, model_out and review) in {path}.')
            if what == 'grad_norm':
                from ..ops import lstm as _lstm
                norm, timeouts = float(host[0][0]), int(host[1][0])
                if _lstm.errors_since_last_report(self._flat.flat.device, timeouts):
                    _lstm.raise_timeout(self._flat.flat.device)
                self._check_other_ranks_loss(host)
                if not np.isfinite(norm):
                    path = self.log_error_state({'state_dict': self.state_dict(), 'optimizer_summary': context})
                    raise RuntimeError(f'The grad_norm ({norm}) is not finite.\n'
                                       f'See error states (model, example, model_out and review) in {path}.')

    @staticmethod
    def _check_other_ranks_loss(host):
        """'flat+words': the third staged value is the sum of ALL ranks' losses.  This rank's own loss has been inspected before (its
        own error); a non-finite sum then means another rank's loss - every rank raises in the same iteration, nobody is left in the
        next collective."""
        if len(host) > 2 and not np.isfinite(float(host[2][0])):
            raise RuntimeError('The loss of another rank is not finite (see its error state).')

    def log_error_state(self, data_dict, folder='log'):
        """trainer.py:640-690: one file per object so a non-picklable one does not lose the rest."""
        log_dir = self.storage_dir / folder
        log_dir.mkdir(parents=True, exist_ok=True)
        for k, v in data_dict.items():
            path = log_dir / f'error_state_{k}.pth'
            try:
                torch.save(v, str(path))
            except Exception as e:  # noqa
                path.with_suffix('.txt').write_text(f'could not be saved: {e!r}')
        return str(log_dir / 'error_state_*.pth')

    # ------------------------------------------------------------------ data parallel
    def enable_bucketed_allreduce(self):
        """Data parallel with ``overlap_allreduce``: build the layer buckets over the flat gradient buffer and hook
        them to gradient completion.  Returns the autograd hook handles (``train`` removes them)."""
        from ..ops import lstm as _lstm
        # (a process group of ONE rank still takes the collective path: that is how the RCCL code is exercised on a
        # single-GPU box, tests/test_gpu_model.py::test_trainer_rccl_path_world_size_1)
        if not self._dp_active() or not self.overlap_allreduce or self._flat is None:
            self._buckets = None
            return []
        self._buckets = buckets = GradBuckets(self.model, self._flat)
        dev = self._flat.flat.device

        def side():
            # the weight-gradient stream of the stream that is current WHEN a gradient becomes ready (the backward pass runs under the
            # forward pass' stream): looked up then, not captured here - train() may be entered under another stream context
            return _lstm._wgrad_stream(dev) if dev.type == 'cuda' else None

        def on_grad(p):
            buckets.ready((p,), side())

        self.op_context.grad_ready_hook = lambda params: buckets.ready(params, side())
        self.op_context.grad_use_hook = buckets.expect
        return [p.register_post_accumulate_grad_hook(on_grad) for p in self._flat.params]

    def _all_ranks_finite(self, mine):
        """Logical AND of ``mine`` over all ranks (one tiny all-reduce)."""
        dev = self._flat.flat.device if self._flat is not None else 'cpu'
        flag = torch.tensor([0. if mine else 1.], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return float(flag.item()) == 0.

    def _broadcast_parameters(self):
        """Step-0 sync: every rank starts from rank 0's weights and buffers."""
        with torch.no_grad():
            for t in itertools.chain(self.model.parameters(), self.model.buffers()):
                dist.broadcast(t, src=0)

    # ------------------------------------------------------------------ checkpoints
    @property
    def checkpoint_dir(self):
        return self.storage_dir / 'checkpoints'

    def default_checkpoint_path(self) -> Path:
        return self.checkpoint_dir / f'ckpt_{self.iteration}.pth'

    def state_dict(self):
        """trainer.py:789-810."""
        return dict(model=self.model.state_dict(), iteration=self.iteration, epoch=self.epoch,
                    optimizer=self.optimizer.state_dict() if self.optimizer.optimizer else None,
                    # the reference's key for hook states (trainer.py:800-810), empty: the fixed hooks of this Trainer keep
                    # their state under their own key, so that the reference Trainer can resume from this file
                    hooks={},
                    ptmi_hooks=dict(best=getattr(self, '_best', None), validation_metric=self.validation_metric,
                                    validation_maximize=getattr(self, 'validation_maximize', False)))

    def load_state_dict(self, state_dict):
        self.model.load_state_dict(state_dict['model'])
        if state_dict.get('optimizer') is not None:
            self.optimizer.load_state_dict(state_dict['optimizer'])
        self.iteration, self.epoch = state_dict['iteration'], state_dict['epoch']
        hooks = state_dict.get('ptmi_hooks')
        if hooks is not None:
            # best-checkpoint tracking continues across a resume - for the metric and direction THIS run registered
            # (register_validation_hook runs before load_checkpoint): a checkpoint written with another metric, another
            # direction or without a validation hook leaves them alone and its `best` value does not carry over
            same = (hooks.get('validation_metric', self.validation_metric) == self.validation_metric
                    and bool(hooks.get('validation_maximize', False)) == bool(getattr(self, 'validation_maximize', False)))
            if same:
                self._best = hooks.get('best')
            elif hooks.get('best') is not None:
                import warnings
                warnings.warn(f"checkpoint tracked its best by {hooks.get('validation_metric')!r} "
                              f"(maximize={hooks.get('validation_maximize')}), this run validates {self.validation_metric!r} "
                              f"(maximize={getattr(self, 'validation_maximize', False)}): starting a new best")
                self._best = None
        # like the reference after a resume (trainer.py:845-851): the triggers have already fired for this iteration
        for trigger in (self.summary_trigger, self.checkpoint_trigger, self.stop_trigger):
            trigger.set_last(self.iteration, self.epoch)

    def save_checkpoint(self, checkpoint_path=None):
        """``ckpt_{iteration}.pth`` + relative symlink ``ckpt_latest.pth`` (trainer.py:812-828)."""
        self._check_pending(flush=True)
        path = Path(checkpoint_path or self.default_checkpoint_path())
        path.parent.mkdir(parents=True, exist_ok=True)
        torch.save(self.state_dict(), str(path))
        latest = path.parent / 'ckpt_latest.pth'
        if latest.is_symlink() or latest.exists():
            latest.unlink()
        latest.symlink_to(path.name)
        best = getattr(self, '_best', None)
        if best is not None and best[1] == self.iteration:
            link = path.parent / f'ckpt_best_{self.validation_metric}.pth'
            if link.is_symlink() or link.exists():
                link.unlink()
            link.symlink_to(path.name)
        return path

    def load_checkpoint(self, map_location='cpu'):
        path = self.checkpoint_dir / 'ckpt_latest.pth'
        assert path.exists(), path
        self.load_state_dict(torch.load(str(path), map_location=map_location, weights_only=False))

    # ------------------------------------------------------------------ runtime test
    def test_run(self, train_iterator, validation_iterator, device=None, deterministic_atol=1e-5,
                 deterministic_rtol=1e-5):
        """Short double training with the invariants of ``runtime_tests.test_run`` (:74-410):
        2 optimizer steps x 2 runs from the same state; validation outputs of both runs agree
        (atol/rtol 1e-5), the first losses agree, training changes the loss and every parameter,
        the review keys are allowed, and the trainer/model state is restored bit-exactly."""
        import copy
        import tempfile
        vmb = self.virtual_minibatch_size
        sub_train = list(itertools.islice(iter(train_iterator), 2 * vmb))
        sub_val = list(itertools.islice(iter(validation_iterator), 2))
        backup = copy.deepcopy(self.state_dict())
        saved = (self.storage_dir, self.iteration, self.epoch, self.stop_trigger, self.checkpoint_trigger,
                 self.summary_trigger, self.validation_iterator, self.summaries)
        saved_validation = (self.validation_metric, getattr(self, 'validation_maximize', False), getattr(self, '_best', None))
        records = []
        try:
            for run in range(2):
                with tempfile.TemporaryDirectory() as tmp:
                    self.optimizer.set_parameters(self.model.parameters())      # fresh optimizer object ...
                    self.load_state_dict(copy.deepcopy(backup))                 # ... then the saved state (Adam moments, step)
                    self.storage_dir = Path(tmp)
                    self.stop_trigger = EndTrigger(2, 'iteration')
                    self.checkpoint_trigger = IntervalTrigger(2, 'iteration')
                    self.summary_trigger = IntervalTrigger(1, 'iteration')
                    self.summaries = []
                    self.register_validation_hook(sub_val)
                    self.train(sub_train, device=device)
                    files = sorted(p.name for p in (Path(tmp) / 'checkpoints').iterdir())
                    assert files == ['ckpt_0.pth', 'ckpt_2.pth', 'ckpt_best_loss.pth', 'ckpt_latest.pth'], files
                    val = [(out, rv) for _, out, rv in self.validate(sub_val)]
                    records.append(dict(summaries=self.summaries, val=val,
                                        params={k: v.detach().clone() for k, v in self.model.named_parameters()}))
            a, b = records
            for (oa, ra), (ob, rb) in zip(a['val'], b['val']):
                for ta, tb_ in zip(oa, ob):
                    torch.testing.assert_close(ta, tb_, atol=deterministic_atol, rtol=deterministic_rtol)
                for k in ra['scalars']:
                    np.testing.assert_allclose(ra['scalars'][k], rb['scalars'][k],
                                               atol=deterministic_atol, rtol=deterministic_rtol)
            first = [[s for s in r['summaries'] if s[1] == 'validation'][0][2]['loss'] for r in records]
            last = [[s for s in r['summaries'] if s[1] == 'validation'][-1][2]['loss'] for r in records]
            np.testing.assert_allclose(first[0], first[1], atol=1e-6)
            assert first[0] != last[0], 'the loss did not change during training'
            self.load_state_dict(copy.deepcopy(backup))
            for k, v in self.model.named_parameters():
                assert not torch.equal(v.detach().cpu(), a['params'][k].cpu()), \
                    f'parameter {k} did not change (zero gradient?)'
        finally:
            self.optimizer.set_parameters(self.model.parameters())
            self.load_state_dict(backup)
            (self.storage_dir, self.iteration, self.epoch, self.stop_trigger, self.checkpoint_trigger,
             self.summary_trigger, self.validation_iterator, self.summaries) = saved
            self.validation_metric, self.validation_maximize, self._best = saved_validation
        for k, v in self.model.state_dict().items():
            assert torch.equal(v.cpu(), backup['model'][k].cpu()), k
        print('Successfully finished test run')
