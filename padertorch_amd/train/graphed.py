"""One optimizer step of a :class:`~padertorch_amd.train.trainer.Trainer` as ONE hipGraph.

The eager step (reference ``padertorch/train/trainer.py:357-393,512-565``: ``train_step`` -> ``backward`` -> ``optimizer_step``) is
~120 launches that the host enqueues in ~4 ms of python; the GPU needs ~6.5 ms for them.  That only hides while the host may run
AHEAD of the GPU, i.e. while nobody looks at a result of the step - but the reference raises a non-finite loss / gradient norm in the
iteration it occurs (``trainer.py:622-636``, ``:740-780``), which needs one host synchronisation per step, after which the GPU waits
for the host at the head of every step (``deferred_checks='step'``: 7.3 instead of 6.8 ms at BASELINE configs[1]).

:class:`GraphedStep` captures the whole step - every micro-step's forward, review, backward, then the clip + Adam + zeroing kernel and
the staged scalars' copies - once per input shape (``torch.cuda.CUDAGraph`` = hipGraph) and replays it: one ``hipGraphLaunch`` per
optimizer step, no python between the kernels.  The loss / gradient-norm / watchdog checks then cost what a synchronisation costs and
raise in the iteration they belong to; the update itself is gated on the device (``csrc/optim.hip``), so a non-finite step leaves the
parameters untouched like the reference's raise in front of ``optimizer.step()``.

What a capture needs from the ops is in ``ops.capture``.  Limits: fixed input shapes (a PackedSequence length pattern is part of the
launch arguments - ragged batches of changing patterns stay on the eager path).

Data parallel (``split_for_allreduce``): with a process group the step is TWO graphs - A: every micro-step's forward + backward and this
rank's two words (sum of its losses, watchdog count); B: norm + clip + Adam + the staged copies - with the exchange between them as
ordinary RCCL calls on the same stream (``Trainer._exchange``: all_reduce(SUM) of the flat gradient bucket, all_reduce(SUM) of the
words; reference ``trainer.py:396-442``).  Nothing of RCCL is captured: a rank that replays and a rank that runs the same step eagerly
(first sighting of a shape) issue the same two collectives, and the summed words give every rank the same update gate and the same
errors in the same iteration.  The bucketed overlap of the eager loop is given up for it: a graph cannot hand a layer's gradients to
a collective outside itself before it ends (cutting it at the bucket boundaries would serialise the weight-gradient queue against the
recurrences at every cut - that costs more than the 0.3-1 ms of an un-overlapped 94 MB all-reduce at W = 8, ``DESIGN.md`` section 5).

What the reference changes BETWEEN iterations is not frozen into the graph: the learning rate, betas, eps, weight decay and the clip
value are device words the optimizer kernel reads (``Adam.refresh_device_hyper``; ``padertorch/train/hooks.py:736,1029`` rewrite
``param_group['lr']``), the loss weights are device words the weighted sum multiplies with (``hooks.py:957-966`` rewrites
``trainer.loss_weights``); every call compares the live values with what the device holds and copies on change.  A weight that moves
between the categories {0, 1, other} changes the launches themselves (``trainer.py:608-620`` skips zero weights; a unit weight has no
kernel here): the step is captured again.
"""
import numpy as np
import torch

from ..ops import capture as _capture

__all__ = ['GraphedStep', 'signature']


def signature(examples):
    """What a captured step bakes in of its examples: structure, tensor shapes / dtypes / devices, and every python NUMBER (lengths,
    frame counts: they become launch arguments).  Two lists of examples with equal signatures can share one graph; ``None`` when
    an example holds a leaf this function does not know (the step then stays eager).  Strings are metadata no kernel reads - the
    reference's batches carry ``example_id`` (``pit/data.py:65``), unique per example - and stay out."""
    import numpy as np
    from ..ops.sequence.pack_module import PaddedList
    parts = []

    def walk(x):
        if torch.is_tensor(x):
            parts.append(('T', tuple(x.shape), str(x.dtype), str(x.device), tuple(x.stride())))
        elif isinstance(x, PaddedList):
            if not x.intact():
                raise TypeError('edited PaddedList')
            parts.append(('PL', tuple(x.padded.shape), str(x.padded.dtype), str(x.padded.device), tuple(x.lengths), bool(x.batch_first)))
        elif isinstance(x, dict):
            parts.append(('D', tuple(x.keys())))
            for v in x.values():
                walk(v)
        elif isinstance(x, (list, tuple)):
            parts.append(('L', type(x).__name__, len(x)))
            for v in x:
                walk(v)
        elif hasattr(x, 'static_tensors') and hasattr(x, 'signature'):
            parts.append(('O',) + tuple(x.signature()))      # (e.g. ops.sequence.StaticSlots: fixed shapes, the pattern is device data)
        elif isinstance(x, str):
            parts.append(('S',))
        elif x is None or isinstance(x, (bool, int, float)):
            parts.append(('V', x))
        elif isinstance(x, np.generic):
            parts.append(('V', x.item()))
        else:
            raise TypeError(type(x).__name__)
    try:
        walk(list(examples))
    except TypeError:
        return None
    return tuple(parts)


class _StaticStage:
    """Pinned host words for the staged scalars of a captured step: the copy nodes of the graph write the same addresses at every replay."""

    def __init__(self, words=1024, device=None):
        self.f32 = torch.empty(words, dtype=torch.float32, pin_memory=True).fill_(float('nan'))
        self.i32 = torch.empty(words, dtype=torch.int32, pin_memory=True).fill_(torch.iinfo(torch.int32).min)
        self.used = {torch.float32: 0, torch.int32: 0}
        self.jobs = []          # [(what, host tensor(s), context, device value(s))] in the order the step staged them
        # loss weights other than 0 and 1 as device words (Trainer._review_to_loss_and_summary multiplies with them while capturing)
        self.lw_keys = []
        self.lw_host = torch.zeros(64, dtype=torch.float32, pin_memory=True)
        self.lw_dev = torch.zeros(64, dtype=torch.float32, device=device) if device is not None else None
        self.lw_held = None

    def loss_weight(self, key, weight):
        """The device word of loss weight ``key`` (a 0-dim view), holding ``weight`` from now on."""
        if key not in self.lw_keys:
            assert len(self.lw_keys) < self.lw_host.numel(), 'too many weighted losses for a captured step'
            self.lw_keys.append(key)
        return self.lw_dev[self.lw_keys.index(key)]

    def refresh_loss_weights(self, loss_weights):
        """Live values of the weighted losses into the device words (a copy on the current stream when one changed)."""
        if not self.lw_keys:
            return
        live = tuple(float(loss_weights[k]) for k in self.lw_keys)
        if live != self.lw_held:
            if self.lw_held is not None:
                torch.cuda.current_stream(self.lw_dev.device).synchronize()      # (the last copy out of these pinned words has run)
            self.lw_host[:len(live)] = torch.tensor(live, dtype=torch.float32)
            self.lw_dev.copy_(self.lw_host, non_blocking=True)
            self.lw_held = live

    def blank(self, shape, dtype):
        assert dtype in self.used, f'staged scalars are fp32 / int32 (got {dtype})'
        n = int(np.prod(shape)) if len(shape) else 1
        buf = self.f32 if dtype == torch.float32 else self.i32
        start = self.used[dtype]
        assert start + n <= buf.numel(), 'too many staged scalars for a captured step'
        self.used[dtype] = start + n
        return buf[start:start + n].view(shape)

    def poison(self):
        self.f32.fill_(float('nan'))
        self.i32.fill_(torch.iinfo(torch.int32).min)


class GraphedStep:
    """``step = GraphedStep(trainer, example_batches)``; ``step(example_batches)`` runs one optimizer step.

    ``examples``: the list of ``virtual_minibatch_size`` device-resident examples of one optimizer step (what ``Trainer.train`` would
    hand to ``train_step`` one by one); their tensors become the graph's static inputs: ``step(new_examples)`` copies new data of the
    same shapes into them (``None`` / the same objects: run on what they hold).
    ``prepare``: optional function ``example -> model input`` captured in front of ``train_step`` (the feature front-end when it is part
    of the step, ``ops.pit_features``).
    The checks run at the end of every call, behind ONE host synchronisation: errors raise in the iteration they belong to, as in the
    reference.  (A mode that inspects them one step late, like the eager ``deferred_checks=True``, was built and measured: 6.591 against
    6.596 ms per step at BASELINE configs[1] - with no python between the launches there is nothing left for the host to run ahead
    with - and removed.)
    """

    def __init__(self, trainer, examples, prepare=None, warmup=2, clone_inputs=False):
        self.trainer = trainer
        self.prepare = prepare
        # ``clone_inputs``: the static inputs are COPIES of the examples' tensors.  Without it the caller's tensors become the static
        # inputs and every later ``load`` overwrites them - fine for a loop that owns its buffers (bench.py), not for a dataset whose
        # device-resident examples come round again in the next epoch (``Trainer.train`` clones)
        self.examples = [self._clone(e) for e in examples] if clone_inputs else list(examples)
        self.device = trainer._flat.flat.device
        assert self.device.type == 'cuda', 'GraphedStep captures a hipGraph: the model has to live on an MI355X'
        for e in self.examples:
            self._strip_records(e)
        self._inputs = [self._tensors(e) for e in self.examples]
        self._stage = None
        self._graph = None
        self._tail = None
        self._words = None
        self._steps = 0
        self._pattern = None
        self.captures = 0
        #: a process group is active: the step is TWO graphs with the data-parallel exchange between them (``split_for_allreduce``)
        self.split = bool(trainer._dp_active()) and trainer.graph_exchange != 'captured'
        #: OPT-IN (``Trainer.graph_exchange = 'captured'``): ONE graph whose nodes include the layer buckets' RCCL all-reduces on the
        #: weight-gradient queue - the eager loop's overlap inside the replay.  RCCL collectives do capture into a hipGraph
        #: (``scripts/mb/rccl_in_graph.py``), but only a group of ONE rank can be formed on the boxes this was built on: with more ranks
        #: it is untested, and every rank has to replay or run eagerly in the same steps (equal shapes on all ranks).
        self.captured_exchange = bool(trainer._dp_active()) and not self.split
        if self.split:
            assert trainer.dp_protocol == 'flat+words' and trainer._buckets is None, \
                "a captured data-parallel step speaks the 'flat+words' protocol (Trainer.dp_protocol; Trainer.train sets it with graph_steps)"
        if self.captured_exchange:
            assert trainer.dp_protocol is None, "graph_exchange = 'captured' keeps the eager loop's collectives (layer buckets, update gate)"
        self.times = None           # split steps: [(graph A, exchange, graph B) in ms of GPU time] of the calls made with record_times
        self._eager(warmup)         # every lazily made table / stream / kernel attribute exists before the capture starts
        self._capture()

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _clone(example):
        from ..ops.sequence.pack_module import PaddedList

        def walk(x):
            if torch.is_tensor(x):
                return x.clone()
            if isinstance(x, PaddedList) and x.intact():
                return PaddedList(x.padded.clone(), x.lengths, x.batch_first, x.lengths_dev)
            if hasattr(x, 'static_tensors') and hasattr(x, 'clone'):
                return x.clone()
            if isinstance(x, dict):
                return type(x)((k, walk(v)) for k, v in x.items())
            if isinstance(x, (list, tuple)):
                return type(x)(walk(v) for v in x)
            return x
        return walk(example)

    @staticmethod
    def _tensors(example):
        """The device tensors of an example in a fixed order: the graph's static inputs (a ``PaddedList`` counts as its ONE padded buffer)."""
        from ..ops.sequence.pack_module import PaddedList
        out = []

        def walk(x):
            if torch.is_tensor(x):
                out.append(x)
            elif isinstance(x, PaddedList) and x.intact():
                out.append(x.padded)
            elif hasattr(x, 'static_tensors'):
                out.extend(x.static_tensors())
            elif isinstance(x, dict):
                for v in x.values():
                    walk(v)
            elif isinstance(x, (list, tuple)):
                for v in x:
                    walk(v)
        walk(example)
        return out

    @staticmethod
    def _strip_records(example):
        """Records that hang on an example's containers and name tensors OUTSIDE the static inputs - the packed log-magnitude a feature
        front-end attaches to ``Y_abs`` (``ops.features.PackedLog1p``: its rows and planes would be those of the captured batch at
        every replay) - are dropped: the model then packs inside the graph."""
        from ..ops.sequence.pack_module import PaddedList

        def walk(x):
            if isinstance(x, PaddedList):
                if getattr(x, 'packed_log1p', None) is not None:
                    x.packed_log1p = None
            elif isinstance(x, dict):
                for v in x.values():
                    walk(v)
            elif isinstance(x, (list, tuple)):
                for v in x:
                    walk(v)
        walk(example)

    def _micro_steps(self):
        """Forward, review and backward of every example of the optimizer step (``trainer.py:357-393``): gradients accumulate."""
        tr = self.trainer
        capturing = _capture.ACTIVE
        if capturing:
            # the head of the graph: ONE fill node zeroes every accumulation word of the step; the parameter-form queue forks here,
            # in front of the front-end kernels (it joins again below: in front of the optimizer kernel that rewrites what it reads)
            from ..ops import lstm as _lstm
            _capture.zero_block(self.device)
            _lstm.begin_captured_step(self.device)
        self._forward_backward()
        if capturing:       # (a failed capture leaves through ops.capture.capture_mode, which forgets the fork)
            _lstm.end_captured_step(self.device)

    def _forward_backward(self):
        tr = self.trainer
        for i, example in enumerate(self.examples):
            if tr._buckets is not None:
                tr._buckets.active = i + 1 == len(self.examples)
            batch = self.prepare(example) if self.prepare is not None else example
            loss, _, _, review = tr.train_step(tr.model, batch, self.device)
            tr.train_summary.update(review)
            tr.backward(loss)
            del loss, review, batch

    def _one_step(self):
        """What ``Trainer.train`` does between two iterations, on the static examples (``trainer.py:357-393,512-532``)."""
        self._micro_steps()
        return self.trainer.optimizer_step()

    def _eager(self, n):
        tr = self.trainer
        keep = tr.deferred_checks
        tr.deferred_checks = True
        try:
            for _ in range(n):
                self._one_step()
            tr._check_pending(flush=True)
        finally:
            tr.deferred_checks = keep
        torch.cuda.synchronize(self.device)

    def _weight_pattern(self):
        """Which loss weights are 0 (term skipped), 1 (no kernel) or anything else (a device word): what the launches depend on."""
        lw = self.trainer.loss_weights
        if lw is None:
            return None
        return tuple((k, 0 if w == 0 else 1 if w == 1 else 2) for k, w in lw.items())

    def _capture(self):
        tr = self.trainer
        assert not tr._dp_active() or self.split or self.captured_exchange
        keep = tr.deferred_checks
        tr._check_pending(flush=True)
        tr.deferred_checks = True               # no host synchronisation inside the capture; this class does the checks
        self._graph = None                      # (a re-capture lets the old graph - and its memory pool - go first)
        self._stage = tr._graph_stage = _StaticStage(device=self.device)
        opt = tr.optimizer
        device_hyper = hasattr(opt, 'refresh_device_hyper') and getattr(opt, '_native_ok', lambda: False)()
        if device_hyper:
            opt.refresh_device_hyper(self.device)
            opt.hyper_from_device = True
        graph = torch.cuda.CUDAGraph()
        running_summary = tr.train_summary
        tr.train_summary = type(running_summary)()      # (the capture's review entries point at the static words: not a step that ran)
        opt_step = tr._opt_step
        self._tail = self._words = None
        try:
            from ..ops import lstm as _lstm
            with _capture.capture_mode():
                if not self.split:
                    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                        self._one_step()
                else:
                    # split_for_allreduce: graph A = every micro-step's forward + backward, joined with the weight-gradient queue, and
                    # this rank's two words; the exchange runs BETWEEN the graphs as ordinary RCCL calls on the same stream
                    # (Trainer._exchange: all ranks issue the same two collectives whether they replay or run the step eagerly);
                    # graph B = norm + clip + Adam (gated by the SUMMED loss word) + the staged scalars' copies.  One memory pool:
                    # what B reads of A (the staged loss values, the words) stays where A left it.
                    # (the words live OUTSIDE the graphs' pool: the collective between the graphs works on ordinary allocations)
                    self._words = torch.zeros(2, dtype=torch.float32, device=self.device)
                    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                        self._micro_steps()
                        _lstm.sync_deferred()
                        self._words.copy_(tr._local_words())
                    self._tail = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._tail, pool=graph.pool(), capture_error_mode='thread_local'):
                        tr._exchanged = self._words
                        tr.optimizer_step()
        finally:
            tr._exchanged = None
            tr._graph_stage = None
            tr.deferred_checks = keep
            tr.train_summary = running_summary
            tr._opt_step = opt_step             # the capture itself executed nothing: not an optimizer step
            if device_hyper:
                opt.hyper_from_device = False
        self._graph = graph
        self._device_hyper = device_hyper
        # what this graph has baked in of the hyper-parameters: where a value CANNOT be a device word (an optimizer off the native path)
        # a change re-captures
        self._baked = None if device_hyper else self._baked_hyper()
        self._pattern = self._weight_pattern()
        self.captures += 1
        # the capture itself executed nothing: parameters, moments, step counts and gradients are what the warm-up left

    def _baked_hyper(self):
        opt = getattr(self.trainer.optimizer, 'optimizer', None)
        groups = getattr(opt, 'param_groups', None) or []
        return (tuple(tuple(sorted((k, repr(v)) for k, v in g.items() if k != 'params')) for g in groups),
                repr(getattr(self.trainer.optimizer, 'gradient_clipping', None)))

    def _refresh(self):
        """Bring everything a replay reads besides its examples up to date; capture again where a change alters the launches."""
        tr = self.trainer
        if self._weight_pattern() != self._pattern or (self._baked is not None and self._baked_hyper() != self._baked):
            self._capture()
        if self._device_hyper:
            tr.optimizer.refresh_device_hyper(self.device)
        if tr.loss_weights is not None:
            self._stage.refresh_loss_weights(tr.loss_weights)

    # ------------------------------------------------------------------ the step
    def load(self, examples):
        """Copy new example data (same structure and shapes) into the graph's static inputs, on the current stream."""
        assert len(examples) == len(self.examples), (len(examples), len(self.examples))
        for example, static in zip(examples, self._inputs):
            new = self._tensors(example)
            assert len(new) == len(static), 'example structure differs from the captured one'
            for src, dst in zip(new, static):
                if src is not dst:
                    assert src.shape == dst.shape and src.dtype == dst.dtype, (src.shape, dst.shape, src.dtype, dst.dtype)
                    dst.copy_(src, non_blocking=True)

    def __call__(self, examples=None, then_load=None):
        """One optimizer step on ``examples`` (``None``: on what the static inputs hold).  ``then_load``: the NEXT step's examples,
        copied into the static inputs right behind this replay and in front of this step's synchronisation - a loop that knows its
        next batch (``data.DevicePrefetcher`` has it on the device already) then starts every step with the replay itself."""
        tr = self.trainer
        self._refresh()
        if examples is not None and examples is not self.examples:
            self.load(examples)
        if not self.split:
            self._graph.replay()
        else:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if self.times is not None else None
            if ev:
                ev[0].record()
            self._graph.replay()
            if ev:
                ev[1].record()
            tr._exchange(self._words)           # in place on graph A's words and on the flat bucket; stream-ordered between the graphs
            if ev:
                ev[2].record()
            self._tail.replay()
            if ev:
                ev[3].record()
        if then_load is not None:
            # (a callable: the next batch is MADE here - host work such as a length pattern's index tables runs while the GPU replays)
            nxt = then_load() if callable(then_load) else then_load
            if nxt is not None:             # (a callable may only PREPARE the next batch - e.g. for another graph's static inputs)
                self.load(nxt)
        tr._opt_step += 1
        self._steps += 1
        # the replay rewrote the parameters through the graph's kernel nodes: nothing bumped their version counters, and the operand
        # forms ops.gemm / ops.lstm cache per (version, pointer) for EAGER forwards between replays - a validation run, test_run, an
        # eager step of another shape - would stay those of the first such forward
        torch.autograd.graph.increment_version(tr._flat.params)
        # ONE synchronisation per optimizer step, behind everything the step consists of; the graph's own copy nodes have left the
        # step's scalars in the static pinned words
        torch.cuda.current_stream(self.device).synchronize()
        if self.split and self.times is not None:
            self.times.append(tuple(ev[i].elapsed_time(ev[i + 1]) for i in range(3)))
        jobs = [(what, host, context) for what, host, context, _ in self._stage.jobs]
        self._inspect(jobs)
        self._record_summary(jobs)
        return self._summary()

    def _record_summary(self, jobs):
        """The step's scalars into the Trainer's running summary as python floats.  The review of the capture names its scalars by
        views into the static words, in staging order: the i-th staged value of a job is the i-th host word of that job."""
        for (what, host, context), (_, static_host, _, _) in zip(jobs, self._stage.jobs):
            scalars = {}
            if what == 'loss':
                # which key of the review reads which static word: compare storage offsets
                base = static_host.data_ptr()
                for key, value in context.get('scalars', {}).items():
                    if torch.is_tensor(value) and value.numel() == 1 and value.is_pinned():
                        i = (value.data_ptr() - base) // value.element_size()
                        if 0 <= i < static_host.numel():
                            scalars[key] = float(host.reshape(-1)[i])
                            continue
                    if not torch.is_tensor(value):
                        scalars[key] = value
                # python values of the capture's review that the run may have changed since: the live ones
                lw = self.trainer.loss_weights
                for key in list(scalars):
                    if lw is not None and key.endswith('_loss_weight') and key[:-len('_loss_weight')] in lw:
                        scalars[key] = lw[key[:-len('_loss_weight')]]
            elif what == 'grad_norm':
                scalars['grad_norm'] = float(host[0][0])
                for key, value in context.get('scalars', {}).items():
                    if not torch.is_tensor(value):
                        scalars[key] = value
                groups = getattr(getattr(self.trainer.optimizer, 'optimizer', None), 'param_groups', None) or []
                for i, group in enumerate(groups):
                    scalars[f'lr/param_group_{i}'] = group['lr']
            self.trainer.train_summary.update({'scalars': scalars})

    def _summary(self):
        out = {'scalars': {}, 'histograms': {}}
        for what, host, _, _ in self._stage.jobs:
            if what == 'grad_norm':
                out['scalars']['grad_norm'] = host[0][0]
                out['histograms']['grad_norm_'] = host[0]
        return out

    def _inspect(self, jobs):
        tr = self.trainer
        from ..ops import lstm as _lstm
        for what, host, context in jobs:
            if what == 'loss':
                value = float(host[-1])
                if not np.isfinite(value):
                    path = tr.log_error_state({'state_dict': tr.state_dict(), 'review': context})
                    raise RuntimeError(f'The loss ({value}) is not finite.\n'
                                       f'See error states (model, example, model_out and review) in {path}.')
            elif what == 'grad_norm':
                norm, timeouts = float(host[0][0]), int(host[1][0])
                if _lstm.errors_since_last_report(self.device, timeouts):
                    _lstm.raise_timeout(self.device)
                tr._check_other_ranks_loss(host)
                if not np.isfinite(norm):
                    path = tr.log_error_state({'state_dict': tr.state_dict(), 'optimizer_summary': context})
                    raise RuntimeError(f'The grad_norm ({norm}) is not finite.\n'
                                       f'See error states (model, example, model_out and review) in {path}.')

    def scalars(self):
        """The staged values of the last replay as python numbers (valid behind a synchronisation: ``checks='step'`` or ``finish``)."""
        out = {}
        for what, host, context, _ in self._stage.jobs:
            if what == 'loss':
                out['loss'] = float(host[-1])
            elif what == 'grad_norm':
                out['grad_norm'] = float(host[0][0])
        return out
