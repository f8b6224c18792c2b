"""`fit` subcommand without Lightning: the training loop the reference delegates to pytorch_lightning.Trainer
(configs/poco.yaml:4-25,60-77; source/poco_model.py:56-132), one process per GPU.

  * optimizer / lr scheduler are instantiated from the YAML `class_path`s (torch.optim.AdamW, MultiStepLR stepped per epoch);
  * trainer.precision: '16-mixed' (fp16 autocast + loss scaling, the reference default), 'bf16-mixed', or '32';
  * every step: device-side batch assembly (data.DeviceBatchLoader) -> model.training_step -> backward -> bucketed gradient
    all-reduce over RCCL overlapped with the backward pass (sharding.GradBuckets) -> optimizer step;
  * BatchNorm / norm_radius buffers follow DDP semantics (`broadcast_buffers=True`): rank 0's values are broadcast, coalesced
    into one message, at the start of every step; between two steps they are rank-local;
  * Adam / AdamW run fused (one launch per dtype group; same update rule);
  * (`PPS_FIT_GRAPH=0` switches it off; with several ranks only forward + backward are captured, the collectives and the optimizer run
    eagerly behind the replay) after three eager steps the WHOLE optimisation step (forward, loss, backward,
    gradient-buffer handling, fused capturable AdamW, loss scaling) is captured into one HIP graph per batch signature and replayed
    (`GraphedStep`); the batch is copied into static buffers; the id tables, their flat forms and CSRs are inputs built by the loader
    (train_graph.table_extras), so the graph holds no sort (replays with the CSR radix sort inside faulted intermittently at the full
    config-3 batch size).  Replay is bit-identical to the eager step (tools/fit_graph_check.py) and ran 1500 consecutive full-size steps
    clean (round-2 probe fit_graph_matrix2.py, git history before round 5); the eager loop spends ~24 ms of Python per step on ~1400 launches and is host-bound (37-40 ms per
    step), the replayed loop is GPU-bound (~30 ms);
  * validation every `check_val_every_n_epoch` epochs in eval() mode -- that is the fused HIP inference path;
  * ModelCheckpoint(save_last) -> models/<name>/version_0/checkpoints/last.ckpt with Lightning's key layout
    ({'state_dict': {'network.<...>': tensor}, 'epoch', 'global_step', 'optimizer_states', 'lr_schedulers'}), so checkpoints
    are interchangeable with the reference; metrics go to models/<name>/version_0/metrics.jsonl.
"""
import contextlib
import importlib
import json
import os
import time

import torch

from . import sharding, train_graph


def _instantiate(spec, *args):
    mod, cls = spec['class_path'].rsplit('.', 1)
    return getattr(importlib.import_module(mod), cls)(*args, **spec.get('init_args', {}))


class _MetricLog:
    """Collects what the model logs through LightningModule.log during a step."""

    def __init__(self):
        self.values = {}

    def __call__(self, name, value, **_kw):
        self.values[name] = value.detach() if torch.is_tensor(value) else float(value)      # tensors are read at the end of the step

    def read(self):
        return {k: (float(v) if torch.is_tensor(v) else v) for k, v in self.values.items()}


def step_stream():
    """A stream of its own for the single-rank optimisation step -- an EXPERIMENT, off by default: `GraphedStep` runs eager steps, the capture and the
    replays on the caller's current stream.  Measured on the config-3 step (profiles/NOTES_r5.md section 3): on a stream of its own 21.3 ms against
    20.25 on the current stream, on a HIGH-priority stream of its own 22.6 (the loader's side stream, which builds the next batch meanwhile, then only
    runs in the gaps and the step waits for its batch).  PPS_STEP_STREAM=1 (and PPS_STEP_PRIORITY=1) select those.  None = the current stream."""
    if not torch.cuda.is_available() or os.environ.get('PPS_STEP_STREAM', '0') != '1':
        return None
    if os.environ.get('PPS_STEP_PRIORITY', '0') != '1':
        return torch.cuda.Stream()
    try:
        hi = int(torch.cuda.Stream.priority_range()[1])
    except Exception:
        hi = -1
    return torch.cuda.Stream(priority=hi)


def _refill(static, batch):
    """The batch into the static input buffers of a recorded step: ONE multi-tensor copy per storage type.  torch._foreach_copy_ only takes its
    one-kernel route for lists of a single dtype -- handed the batch's 68 tensors (int64 tables, fp32 points, uint8 / bool masks) as one list
    it issues 68 device memcpys, 0.27 ms of the step's queue between the optimizer of one step and the replay of the next
    (tools/dbg/copybuffer_queues.sh, profiles/NOTES_r6.md section 9)."""
    groups = {}
    for k, dst in static.items():
        pair = groups.setdefault(dst.dtype, ([], []))
        pair[0].append(dst)
        pair[1].append(batch[k])
    for dsts, srcs in groups.values():
        torch._foreach_copy_(dsts, srcs)


class GraphedStep:
    """One optimisation step as a replayable HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm), one graph per batch signature
    (keys, shapes, dtypes).  `eager(batch, bi)` is the step body; it must be free of host synchronisation and data-dependent shapes
    (it is: metrics are device tensors, id tables are inputs, the optimizer is fused + capturable)."""
    WARMUP = 3

    def __init__(self, eager, metrics, enabled=True, max_graphs=2, after_capture=None, after_replay=None, on_capture_failed=None):
        self.eager, self.metrics, self.enabled, self.max_graphs = eager, metrics, enabled, max_graphs
        self.after_capture, self.after_replay = after_capture, after_replay      # host-side state a replay cannot reproduce (hooks that fired)
        self.on_capture_failed = on_capture_failed      # puts host-side state a half-recorded step left behind (loss scaler stage, gradient buckets) back
        self.seen, self.graphs, self.failed, self._done = {}, {}, False, None
        self.replayed = False
        self.stream = step_stream()

    def touch(self, module):
        """A replayed graph updates parameters and buffers in place WITHOUT the tensors' version counters moving (no op is dispatched), and
        the inference plans of ppsurf_amd.modules are cached on those counters: before anything reads the module through eval()
        (validation, predict after fit), bump them with an exact no-op."""
        if not self.replayed:
            return
        self.replayed = False
        with torch.no_grad():
            ts = list(module.parameters()) + list(module.buffers())
            fl = [t for t in ts if t.is_floating_point()]
            it = [t for t in ts if not t.is_floating_point()]
            if fl:
                torch._foreach_mul_(fl, 1.0)
            if it:
                torch._foreach_add_(it, 0)

    @staticmethod
    def signature(batch):
        """What a recorded step is valid for: names, shapes and types of the tensors, and the VALUES of scalar entries (a number / string / bool a
        step may branch on is baked into its graph).  List entries (`pc_file_in`: file names, different for every batch) are neither part of the
        signature nor handed to the recorded step -- a step that read them would fail at capture time instead of silently replaying stale ones."""
        sig = [(k, tuple(v.shape), str(v.dtype)) for k, v in batch.items() if torch.is_tensor(v)]
        sig += [(k, 'value', repr(v)) for k, v in batch.items() if isinstance(v, (int, float, bool, str)) and not k.startswith('_')]
        return tuple(sorted(sig))

    def run(self, batch, bi):
        """Executes the step (eagerly, or by replaying the captured graph) and leaves the logged values in self.metrics.values; on the caller's
        current stream unless step_stream() hands out one of its own."""
        if self.stream is None:
            return self._run(batch, bi)
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self._run(batch, bi)
        cur.wait_stream(self.stream)

    def _run(self, batch, bi):
        if self.enabled and not self.failed:
            sig = self.signature(batch)
            entry = self.graphs.get(sig)
            if entry is None:
                n = self.seen.get(sig, 0)
                self.seen[sig] = n + 1
                if n >= self.WARMUP and len(self.graphs) < self.max_graphs:
                    entry = self._capture(batch, bi)           # records only; the replay below executes this batch
                    if entry is not None:
                        self.graphs[sig] = entry
            if entry is not None:
                static, graph, logged, state = entry
                # The static inputs may only be overwritten once the previous replay has finished READING them.  Stream order should
                # guarantee that, but with the host several steps ahead the copies were observed to race the tail of the previous
                # replay (id tables changing under the CSR sort -> out-of-bounds scatter inside rocprim's onesweep kernel,
                # round-2 probe fit_graph_matrix2.py, git history); waiting on an event recorded behind the replay costs nothing -- the id-table kernels
                # of this step are already queued behind the previous replay when the host gets here.
                if self._done is not None:
                    self._done.synchronize()
                _refill(static, batch)
                graph.replay()
                self.replayed = True
                if self._done is None:
                    self._done = torch.cuda.Event()
                self._done.record()
                # the logged tensors are outputs of the graph and are overwritten by the next replay: hand out copies
                self.metrics.values = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in logged.items()}
                if self.after_replay is not None:
                    self.after_replay(state)
                return
        self.eager(batch, bi)

    def _capture(self, batch, bi):
        static = {k: v.clone() for k, v in batch.items() if torch.is_tensor(v)}
        # scalar entries are part of the signature and may be read; lists (file names) and '_...' caches of device tensors are withheld
        rest = {k: v for k, v in batch.items() if isinstance(v, (int, float, bool, str)) and not k.startswith('_')}
        graph = torch.cuda.CUDAGraph()
        try:
            torch.cuda.synchronize()
            with _no_gc_during_capture():
                with torch.cuda.graph(graph, stream=self.stream, capture_error_mode='thread_local'):
                    self.eager(dict(static, **rest), bi)
            logged = dict(self.metrics.values)
        except Exception as exc:                               # anything that cannot be captured: stay eager for the rest of the run
            self.failed = True
            if os.environ.get('PPS_FIT_GRAPH_DEBUG'):
                import traceback
                traceback.print_exc()
            print('fit: HIP-graph capture of the step failed ({}: {}); continuing eagerly'.format(type(exc).__name__, str(exc).split('\n')[0]))
            torch.cuda.synchronize()
            if self.on_capture_failed is not None:
                self.on_capture_failed()                        # the eager re-run of this step starts from clean host-side state
            return None
        return static, graph, logged, (self.after_capture() if self.after_capture is not None else None)


class _no_gc_during_capture:
    """Old cyclic garbage is collected BEFORE a recording and the collector is off during it: what a collection frees in the middle of a stream
    capture -- recorded graphs, streams, device tensors used on other streams, left behind by whatever ran in the process before (a validation pass,
    a reconstruction, another fit) -- is destroyed with the capture open, and the recording then ended with `capturing stream has unjoined work`
    (seen as the last test of the GPU suite falling back to the eager step after 470 other tests, never in a fresh process)."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.collect()
        gc.disable()
        return self

    def __exit__(self, *exc):
        if self.was:
            import gc
            gc.enable()
        return False


class StagedStep:
    """Forward + backward of a MULTI-RANK step with the gradient all-reduce overlapped with the backward pass -- eagerly and when the step is
    replayed from HIP graphs.  The reference trains under Lightning DDP (configs/device_server.yaml:2, source/base/mp.py:85-91), whose autograd
    hooks start a bucket's all-reduce while backward continues; hooks do not fire in a replayed graph, and a collective cannot be recorded into one
    here.  So the backward pass is cut into train_graph.N_STAGES pieces (train_graph.BackwardStages: behind the encoder's coarse levels and behind
    its middle levels), bucket k of sharding.GradBuckets holds exactly the parameters stage k completes, every stage is recorded as its OWN graph
    (one memory pool, always replayed in the same order) and the loop is

        replay(graph 0: zero, forward, loss, backward stage 0, pack bucket 0)   ->  all-reduce(bucket 0) issued, asynchronous
        replay(graph 1: backward stage 1, pack bucket 1)                        ->  all-reduce(bucket 1)      (bucket 0 is on the wire meanwhile)
        replay(graph 2: backward stage 2, pack bucket 2)                        ->  all-reduce(bucket 2)
        finish(): wait, average, the per-parameter mask collective; optimizer step (eager, behind)

    Bucket 0 is 82 % of the gradient bytes and is complete when ~70 % of the backward pass is still to run.  Eager steps (warm-up, a failed
    capture) run the same stages and issue the same collectives in the same order, so ranks that replay and ranks that do not stay in step."""
    WARMUP = 3

    def __init__(self, model, buckets, scaler, ctx, metrics, enabled=True, max_graphs=2, on_capture_failed=None):
        self.model, self.buckets, self.scaler, self.ctx, self.metrics = model, buckets, scaler, ctx, metrics
        self.enabled, self.max_graphs, self.on_capture_failed = enabled, max_graphs, on_capture_failed
        self.seen, self.graphs, self.failed, self._done, self.replayed = {}, {}, False, None, False
        self.staged = len(buckets.buckets) == train_graph.N_STAGES      # else: one backward pass, every collective in finish()
        if not self.staged and sharding.world()[0] == 0:
            print('fit: {} gradient buckets for {} backward stages -- the step runs as ONE backward pass, eagerly, with every all-reduce behind it '
                  '(no HIP-graph replay, no all-reduce / backward overlap); build the buckets from train_graph.parameter_stages(model)'.format(
                      len(buckets.buckets), train_graph.N_STAGES))
        # ONE stream for the eager steps, the captures and the replays: autograd binds a parameter's gradient accumulation to the stream of the
        # parameter's first use, and an accumulator that survives from an eager step on another stream would pull that stream into the capture
        # as an unjoined branch (hipStreamEndCapture faults on it; measured, profiles/NOTES_r5.md)
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None

    touch = GraphedStep.touch

    def _log(self, what):
        if self.buckets.order_log is not None:
            self.buckets.order_log.append(what)

    def _forward(self, batch, bi):
        self.buckets.zero()
        self.metrics.values = {}
        with self.ctx:
            # a COPY of the dictionary: the networks add entries to it (`latents` carries the step's autograd graph, ppsurf_model.py:70-117), and a
            # caller that keeps its batch would keep that graph -- and the parameters' gradient accumulators with their stream -- alive into the
            # next step's capture
            loss = self.model.training_step(dict(batch), bi)
        return self.scaler.scale(loss)

    def eager(self, batch, bi):
        if not self.staged:
            self._forward(batch, bi).backward()
            self.model.on_after_backward()
            return
        with train_graph.staged() as st:
            scaled = self._forward(batch, bi)

            def after(k):
                self._log('stage{}'.format(k))
                self.buckets.reduce(k)
            st.backward(scaled, after)
        self.model.on_after_backward()

    def run(self, batch, bi):
        if self.stream is None:
            return self._run(batch, bi)
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self._run(batch, bi)
        cur.wait_stream(self.stream)

    def _run(self, batch, bi):
        if self.enabled and not self.failed and self.staged:
            sig = GraphedStep.signature(batch)
            entry = self.graphs.get(sig)
            if entry is None:
                n = self.seen.get(sig, 0)
                self.seen[sig] = n + 1
                if n >= self.WARMUP and len(self.graphs) < self.max_graphs:
                    entry = self._capture(batch, bi)
                    if entry is not None:
                        self.graphs[sig] = entry
            if entry is not None:
                static, graphs, logged, touched = entry
                if self._done is not None:
                    self._done.synchronize()                     # the previous replay has finished reading the static inputs (GraphedStep.run)
                _refill(static, batch)
                self.buckets.begin_replay(touched)
                for k, g in enumerate(graphs):
                    g.replay()
                    self._log('replay{}'.format(k))
                    self.buckets.reduce(k)                       # asynchronous: on the wire while the next sub-graph runs
                self.replayed = True
                if self._done is None:
                    self._done = torch.cuda.Event()
                self._done.record()
                self.metrics.values = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in logged.items()}
                return
        self.eager(batch, bi)

    def _capture(self, batch, bi):
        static = {k: v.clone() for k, v in batch.items() if torch.is_tensor(v)}
        rest = {k: v for k, v in batch.items() if isinstance(v, (int, float, bool, str)) and not k.startswith('_')}
        graphs = [torch.cuda.CUDAGraph() for _ in range(train_graph.N_STAGES)]
        try:
            torch.cuda.synchronize()
            with _no_gc_during_capture(), train_graph.staged() as st:
                with torch.cuda.graph(graphs[0], stream=self.stream, capture_error_mode='thread_local'):
                    scaled = self._forward(dict(static, **rest), bi)
                    st.run_stage(0, scaled)
                    self.buckets.pack(0)
                assert st.n_stages == train_graph.N_STAGES
                for k in range(1, train_graph.N_STAGES):
                    # the later stages read what graph 0 left in ITS memory (saved activations, the cuts' gradients): one pool, fixed replay order
                    with torch.cuda.graph(graphs[k], pool=graphs[0].pool(), stream=self.stream, capture_error_mode='thread_local'):
                        st.run_stage(k)
                        self.buckets.pack(k)
                del scaled
                st.release()
            self.model.on_after_backward()
            logged = dict(self.metrics.values)
        except Exception as exc:
            self.failed = True
            if os.environ.get('PPS_FIT_GRAPH_DEBUG'):
                import traceback
                traceback.print_exc()
            print('fit: HIP-graph capture of the staged step failed ({}: {}); continuing eagerly'.format(type(exc).__name__, str(exc).split('\n')[0]))
            torch.cuda.synchronize()
            if self.on_capture_failed is not None:
                self.on_capture_failed()
            return None
        return static, graphs, logged, set(self.buckets.touched)


class HostGcPacer:
    """Python's cyclic garbage collector paused for the duration of a step loop.  Every step allocates a few hundred tensor / dict / tuple objects, so
    the collector's oldest generation comes due every ~80 steps and walks every live object of the process (all modules, parameters, cached plans):
    110-185 ms during which nothing is launched -- and the fit loop waits for the previous replay before it refills the static inputs, so the GPU
    idles for all of it (tools/fit_step_jitter.py: 23.1 -> 21.0 ms per step averaged over 240 steps; a reconstruction loses 30-45 ms of its 0.5 s
    the same way).  Reference counting still frees everything that is not part of a cycle at once; `tick()` runs a young-generation collection every
    `every` steps (sub-millisecond), the full collection happens when the loop ends (`close()` / leaving the `with`).
    Opt out with PPS_HOST_GC=auto."""

    FULL_EVERY = 16                   # every FULL_EVERY young collections (1024 steps at the default) one full collection: objects promoted to the
                                      # oldest generation (eager steps: autograd closures, tracebacks, loader dicts holding device tensors) would
                                      # otherwise pile up until the epoch ends; ~150 ms per 1024 steps amortised

    def __init__(self, every=64):
        self.every, self.n, self.active = int(every), 0, False

    def __enter__(self):
        import gc
        if os.environ.get('PPS_HOST_GC', 'paced') != 'auto' and gc.isenabled():
            gc.collect()
            gc.freeze()                   # what is alive now (model, plans, loaders) is never walked again until close()
            gc.disable()
            self.active = True
        return self

    def tick(self):
        self.n += 1
        if self.active and self.n % self.every == 0:
            import gc
            gc.collect(2 if (self.n // self.every) % self.FULL_EVERY == 0 else 1)

    def close(self):
        if self.active:
            import gc
            self.active = False
            gc.enable()
            gc.unfreeze()
            gc.collect()

    def __exit__(self, *exc):
        self.close()
        return False


def autocast_context(precision, device_type='cuda'):
    precision = str(precision)
    if precision in ('16-mixed', '16'):
        return torch.autocast(device_type, dtype=torch.float16), True
    if precision in ('bf16-mixed', 'bf16'):
        return torch.autocast(device_type, dtype=torch.bfloat16), False
    if precision in ('32', '32-true'):
        return contextlib.nullcontext(), False
    raise ValueError('unsupported trainer.precision {!r}'.format(precision))


LIGHTNING_CKPT_VERSION = '2.0.0'


def save_checkpoint(path, model, optimizer, scheduler, epoch, global_step):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = path + '.tmp'
    # key layout of a Lightning 2 checkpoint (the reference pins pytorch-lightning>=2.0, requirements.txt:3): the version must be
    # a valid PEP 440 string (Lightning's migrate_checkpoint parses it), 'loops' / 'callbacks' may be empty
    osd = optimizer.state_dict()
    # plain floats, whatever the run kept on the device; with graph replay the learning rate lives in a float32 device tensor, so the exact
    # double is taken from the scheduler's closed form where it has one (MultiStepLR: base_lr * gamma ** milestones passed)
    lrs = [float(g['lr']) for g in osd['param_groups']]
    if scheduler is not None and hasattr(scheduler, '_get_closed_form_lr'):
        try:
            closed = [float(v) for v in scheduler._get_closed_form_lr()]
            if len(closed) == len(lrs) and all(abs(a - b) <= 1e-6 * abs(b) for a, b in zip(closed, lrs)):
                lrs = closed
        except Exception:
            pass
    osd['param_groups'] = [dict(g, lr=lr) for g, lr in zip(osd['param_groups'], lrs)]
    torch.save({'state_dict': model.state_dict(), 'epoch': epoch, 'global_step': global_step,
                'optimizer_states': [osd], 'lr_schedulers': [scheduler.state_dict()] if scheduler is not None else [],
                'pytorch-lightning_version': LIGHTNING_CKPT_VERSION, 'loops': {}, 'callbacks': {}, 'hyper_parameters': {},
                'ppsurf_amd': True}, tmp)
    os.replace(tmp, path)


def fit(model, data, cfg, ckpt_path=None, device='cuda', log=print):
    rank, world = sharding.world()
    multi = sharding.multi()                                   # several ranks -- or one rank made to run the multi-rank step (PPS_SINGLE_RANK_COLLECTIVES)
    tcfg = cfg.get('trainer', {})
    max_epochs = int(tcfg.get('max_epochs', 150))
    max_steps = int(tcfg.get('max_steps', -1))
    val_every = int(tcfg.get('check_val_every_n_epoch', 1))
    ctx, use_scaler = autocast_context(tcfg.get('precision', '16-mixed'), torch.device(device).type)
    params = [p for p in model.parameters() if p.requires_grad]
    if not cfg.get('optimizer'):
        raise ValueError('fit needs an `optimizer:` section (class_path / init_args), as in configs/poco.yaml:60-69')
    on_gpu = torch.device(device).type == 'cuda'
    graph_on = on_gpu and os.environ.get('PPS_FIT_GRAPH', '1') != '0'
    use_graph = graph_on and not multi                         # the whole step (optimizer included) as one graph
    split_graph = graph_on and multi                           # forward + backward as a graph; collectives and optimizer eager behind it
    ospec = cfg['optimizer']
    if on_gpu and ospec.get('class_path', '').rsplit('.', 1)[-1] in ('AdamW', 'Adam') and 'fused' not in ospec.get('init_args', {}):
        # the fused implementation (one launch per dtype group instead of ~10 small foreach launches over 298 parameter tensors: 57.6 ->
        # 46.6 ms per step) applies the same update rule, takes the loss scale / found-inf tensors on the device (no host sync in
        # GradScaler.step) and, with capturable=True, keeps its step counter on the device so that the step can be recorded into a graph
        ospec = dict(ospec, init_args=dict(ospec.get('init_args', {}), fused=True, capturable=use_graph))
        if ospec['class_path'] == 'torch.optim.AdamW':
            # the same optimizer (state, checkpoints, GradScaler and graph-capture behaviour of the fused torch class it derives from) whose step is
            # ONE launch over a table of 4096-element pieces instead of torch's 65 536-element slabs: 0.78 -> 0.15 ms per step (ppsurf_amd/optim.py)
            ospec = dict(ospec, class_path='ppsurf_amd.optim.AdamW')
    optimizer = _instantiate(ospec, params)

    scheduler = _instantiate(cfg['lr_scheduler'], optimizer) if cfg.get('lr_scheduler') else None
    scaler = torch.amp.GradScaler(torch.device(device).type, enabled=use_scaler)
    start_epoch, global_step = 0, 0
    if ckpt_path is not None:
        state = torch.load(ckpt_path, map_location='cpu')
        model.load_state_dict(state['state_dict'])
        if state.get('optimizer_states'):
            optimizer.load_state_dict(state['optimizer_states'][0])
        if scheduler is not None and state.get('lr_schedulers'):
            scheduler.load_state_dict(state['lr_schedulers'][0])
        start_epoch, global_step = int(state.get('epoch', -1)) + 1, int(state.get('global_step', 0))
    if use_graph:
        for group in optimizer.param_groups:                   # a device-side learning rate: the scheduler's changes reach the replayed graph
            group['lr'] = torch.tensor(float(group['lr']), dtype=torch.float32, device=device)
    if multi:
        import torch.distributed as dist
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)
    # several ranks: bucket k = the parameters backward stage k completes (train_graph.parameter_stages), all-reduced while stage k + 1 runs
    buckets = sharding.GradBuckets(params, defer=multi, groups=train_graph.parameter_stages(model) if multi else None)
    out_dir = os.path.join('models', str(getattr(model, 'name', 'model')), 'version_0')
    ckpt_file = os.path.join(out_dir, 'checkpoints', 'last.ckpt')
    metrics = _MetricLog()
    model.__dict__['_fit_log'] = metrics
    train_loader, val_loader = data.train_dataloader(), data.val_dataloader()
    train_loader.thread_collate = use_graph or split_graph      # see data.DeviceBatchLoader.__iter__
    mfile = None
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        mfile = open(os.path.join(out_dir, 'metrics.jsonl'), 'a')
    history = []
    done = False
    pending = None

    def flush(item):
        if item is None:
            return
        values, extra = item
        rec = dict({k: (float(v) if torch.is_tensor(v) else v) for k, v in values.items()}, **extra)
        history.append(rec)
        if mfile is not None:
            mfile.write(json.dumps(rec) + '\n')

    def reset_host_state():
        """After a capture that failed part-way: the loss scaler may be left in its 'unscaled' / 'stepped' stage for this optimizer and the gradient
        buckets half filled; the step is re-run eagerly right after, from a clean slate."""
        buckets.zero()
        per_opt = getattr(scaler, '_per_optimizer_states', None)
        if per_opt is not None:
            per_opt.clear()

    def eager_step(batch, bi):                                 # one rank: the whole step, what GraphedStep records and replays
        compute_step(batch, bi)
        apply_step()

    def compute_step(batch, bi):
        buckets.zero()
        metrics.values = {}
        with ctx:
            loss = model.training_step(batch, bi)
        scaler.scale(loss).backward()
        model.on_after_backward()

    def apply_step():
        buckets.finish()
        scaler.step(optimizer)
        scaler.update()
        train_graph.release_step_caches()

    if multi:
        # several ranks: forward + backward in stages (eager or as one replayed HIP graph per stage) with the bucket all-reduces issued between the
        # stages (StagedStep); buffer broadcast, the mask collective and the optimizer run eagerly around it
        core = StagedStep(model, buckets, scaler, ctx, metrics, enabled=split_graph, on_capture_failed=reset_host_state)
        if os.environ.get('PPS_FIT_ORDER_LOG'):
            buckets.order_log = []

        class _Split:
            graphs, failed = core.graphs, False

            @staticmethod
            def run(batch, bi):
                sharding.broadcast_buffers(model)
                core.run(batch, bi)
                _Split.failed = core.failed
                apply_step()

            touch = staticmethod(core.touch)
        stepper = _Split
    else:
        # PPS_FIT_GRAPH=norecord (tests): everything the graph mode sets up -- capturable optimizer, device-side learning rate, batches built by the
        # loader thread -- but the step is never recorded: the eager twin a replayed fit is compared with
        stepper = GraphedStep(eager_step, metrics, enabled=use_graph and os.environ.get('PPS_FIT_GRAPH') != 'norecord',
                              on_capture_failed=reset_host_state)
    pacer = HostGcPacer()
    for epoch in range(start_epoch, max_epochs):
        host_lr = float(optimizer.param_groups[0]['lr'])        # once per epoch (the scheduler steps per epoch): no per-step read of a device value
        model.train()
        train_loader.set_epoch(epoch)
        t0 = time.time()
        with pacer:                                            # no stop-the-world collection inside the step loop (HostGcPacer); the full
            for bi, batch in enumerate(train_loader):          # collection of the epoch runs when the block is left, also by an exception
                stepper.run(batch, bi)
                pacer.tick()
                global_step += 1
                # the step's logged values are still device tensors: they are read one step LATER (when they are long finished), so
                # the host never waits for the GPU inside the loop and keeps queueing the next step's launches
                flush(pending)
                pending = (metrics.values, dict(epoch=epoch, step=global_step, lr=host_lr))
                if 0 < max_steps <= global_step:
                    done = True
                    break
            flush(pending)
            pending = None
        buckets.check()                                        # world > 1: the set of parameters with gradients never changed (also for epochs shorter than
                                                               # GradBuckets.CHECK_EVERY steps, and before the checkpoint below is written)
        stepper.touch(model)                                   # replays move no version counters: the eval() plans below are keyed on them
        if scheduler is not None:
            scheduler.step()
        last = history[-1] if history else {}
        msg = 'epoch {} ({} steps, {:.2f} s): train loss {:.4f}'.format(epoch, global_step, time.time() - t0,
                                                                      last.get('loss/train/00_all', float('nan')))
        if val_every > 0 and (epoch + 1) % val_every == 0 and len(val_loader.dataset) > 0:
            model.eval()
            metrics.values = {}
            vals = []
            with torch.no_grad():
                for bi, batch in enumerate(val_loader):
                    vals.append(float(model.validation_step(batch, bi)))
            vloss = sharding.weighted_mean_over_ranks(sum(vals), len(vals), device)      # ranks weigh in by their batch count
            msg += ', val loss {:.4f}'.format(vloss)
            if mfile is not None:
                mfile.write(json.dumps({'epoch': epoch, 'loss/val/00_all': vloss, 'metrics/val/F1': metrics.read().get('metrics/val/F1')}) + '\n')
        if rank == 0:
            save_checkpoint(ckpt_file, model, optimizer, scheduler, epoch, global_step)
            mfile.flush()
            log(msg)
        if done:
            break
    if mfile is not None:
        mfile.close()
    if buckets.order_log is not None:                          # PPS_FIT_ORDER_LOG=1 (tests): the host order of stage replays and bucket collectives
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'order_rank{}.json'.format(rank)), 'w') as f:
            json.dump(buckets.order_log, f)
    model.__dict__.pop('_fit_log', None)
    if (use_graph or split_graph) and rank == 0:
        print('fit: HIP-graph replay of the step: {} graph(s) captured{}'.format(len(stepper.graphs), ', capture FAILED (ran eagerly)' if stepper.failed else ''))
    return history
