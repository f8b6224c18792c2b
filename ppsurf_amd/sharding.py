"""Query-block sharding across the GPUs of one node (one process per GPU, torch.distributed; backend 'nccl' = RCCL over
xGMI on the GPU box, 'gloo' in the CPU tests).

Queries are independent given the replicated per-shape state (cloud + per-point table, ~104 MB), so the data path needs a
single exchange per growth round: a variable-length all-gather of 4 bytes per query."""
import contextlib
import os

import torch


_QUERY_SHARDING = False


def set_query_sharding(on: bool):
    """Query-block sharding inside one shape is opt-in: with shape-level sharding the ranks work on different shapes and
    must not meet in a collective."""
    global _QUERY_SHARDING
    _QUERY_SHARDING = bool(on)


def world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def single_rank_collectives() -> bool:
    """PPS_SINGLE_RANK_COLLECTIVES=1: a process group of ONE rank still runs every collective of the multi-rank code paths (query-sharded predict,
    staged data-parallel fit).  RCCL accepts a one-rank communicator, so on a 1-GPU box every collective call site -- the padded all-gather of
    sharded_map, the latent all-reduce, the bucket all-reduces issued between the replayed backward stages, the buffer broadcast, the mask MAX --
    executes on the REAL backend with its dtype / stream semantics (tests/test_gpu_nccl_single_rank.py, `bench.py --single-rank-collectives`)."""
    return os.environ.get('PPS_SINGLE_RANK_COLLECTIVES', '0') == '1'


def multi() -> bool:
    """True when the data path issues its collectives: an initialised process group of several ranks, or of one rank under
    PPS_SINGLE_RANK_COLLECTIVES=1."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or single_rank_collectives()


def init_process_group(device=None, backend=None):
    """One process per GPU: the group of the launcher's RANK / WORLD_SIZE (torch.distributed.run) on 127.0.0.1 unless MASTER_ADDR says otherwise;
    backend PPS_BACKEND (default 'nccl' = RCCL over xGMI; 'gloo' for rehearsals on one GPU).  With RCCL the communicator is bound to `device`
    at once (device_id), so the first collective does not have to guess it.  No-op when a group exists; returns (rank, world)."""
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    rank, ws = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    backend = backend or os.environ.get('PPS_BACKEND', 'nccl')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if 'MASTER_PORT' not in os.environ:
        if ws > 1:
            raise RuntimeError('WORLD_SIZE > 1 without MASTER_PORT: launch the ranks with python -m torch.distributed.run')
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(('127.0.0.1', 0))
            os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
    if backend == 'nccl' and device is not None and torch.device(device).type == 'cuda':
        dist.init_process_group('nccl', rank=rank, world_size=ws, device_id=torch.device(device))
    else:
        dist.init_process_group(backend, rank=rank, world_size=ws)
    return rank, ws


def shard_range(n: int, rank: int, world_size: int):
    """Contiguous, balanced [lo, hi) slice of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# a rank never gets fewer queries than this: the persistent decoder kernels need ~8k queries to fill 256 CUs.  PPS_MIN_SHARD overrides it (the first
# runs on an 8-GPU node can sweep it against the real all-gather latency over xGMI without touching the code)
MIN_SHARD = int(os.environ.get('PPS_MIN_SHARD', '8192'))
STATS = {'collective_events': None, 'calls': 0, 'items': 0}


def profile_collectives(on: bool):
    """Record a (start, end) CUDA event pair around every data-path collective (read with collective_seconds())."""
    STATS['collective_events'] = [] if on else None
    STATS['calls'], STATS['items'] = 0, 0


def collective_seconds() -> float:
    ev = STATS['collective_events'] or []
    if ev:
        torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) * 1e-3


@contextlib.contextmanager
def _timed_collective(device):
    ev = STATS['collective_events']
    if ev is None or torch.device(device).type != 'cuda':
        yield
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    yield
    b.record()
    ev.append((a, b))


def shard_ranges(n: int, world_size: int, min_shard: int = None):
    """[lo, hi) of every rank: the first `active` ranks share the n items evenly, where `active` is the largest rank count that
    keeps every share >= min_shard (at least one rank); the others get empty ranges."""
    min_shard = MIN_SHARD if min_shard is None else min_shard
    active = max(1, min(world_size, n // max(1, min_shard)))
    out = [shard_range(n, r, active) if r < active else (n, n) for r in range(world_size)]
    return out


def sharded_map(fn, items: torch.Tensor, min_shard: int = None) -> torch.Tensor:
    """Every rank evaluates fn on its contiguous slice of `items` [n, ...] (fn returns one float32 per item) and all
    ranks receive the concatenated result [n] in the original order: one all-gather of 4 bytes per item.
    Small lists are not split below MIN_SHARD items per rank (ranks without a share only take part in the collective)."""
    import torch.distributed as dist
    rank, ws = world()
    n = items.shape[0]
    if not multi() or not _QUERY_SHARDING:
        return fn(items)
    ranges = shard_ranges(n, ws, min_shard)
    lo, hi = ranges[rank]
    width = max(h - l for l, h in ranges)
    pad = torch.zeros((max(width, 1),), dtype=torch.float32, device=items.device)
    if hi > lo:
        pad[:hi - lo] = fn(items[lo:hi]).to(torch.float32)
    parts = [torch.empty_like(pad) for _ in range(ws)]
    with _timed_collective(items.device):
        dist.all_gather(parts, pad)
    STATS['calls'] += 1
    STATS['items'] += n
    return torch.cat([parts[r][:h - l] for r, (l, h) in enumerate(ranges)])


def max_over_ranks(seconds: float, device) -> float:
    import torch.distributed as dist
    if not multi():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def mean_over_ranks(value: float, device) -> float:
    import torch.distributed as dist
    _, ws = world()
    if not multi():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / ws


def weighted_mean_over_ranks(total: float, count: int, device) -> float:
    """sum(total over ranks) / sum(count over ranks): a rank without batches contributes nothing instead of a zero."""
    import torch.distributed as dist
    if multi():
        t = torch.tensor([total, float(count)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total, count = float(t[0]), float(t[1])
    return total / count if count > 0 else float('nan')


def allreduce_latents(latent_sum: torch.Tensor, counts: torch.Tensor):
    """Sum the per-rank partial latent sums / counts of a latent loop whose encoder passes were dealt round-robin."""
    import torch.distributed as dist
    if multi():
        with _timed_collective(latent_sum.device):
            dist.all_reduce(latent_sum, op=dist.ReduceOp.SUM)
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return latent_sum, counts


_BUFFER_PLANS = {}


def broadcast_buffers(module, src: int = 0):
    """torch DDP's `broadcast_buffers=True` (the Lightning default): at the start of every step the floating-point buffers
    (BatchNorm running statistics, FKAConv norm_radius -- the latter is read by the train-mode forward) are overwritten with
    rank `src`'s values, coalesced into ONE broadcast.
    Three launches per step: a persistent flat fp32 buffer with one view per module buffer is kept per module (rebuilt if a buffer moved); pack
    (one multi-tensor copy), broadcast, unpack (one multi-tensor copy).  Until round 6 every step built the flat tensor anew -- ~150 reshape / cast
    calls and a 150-input cat -- and copied back buffer by buffer: 3.3-5.0 ms of host time per multi-rank step.  (The host runs ahead of the
    GPU, so the step time did not move: 24.4 ms per staged step at B = 10 on one rank before and after, against 19.8-20.1 for the single-graph
    step -- the staged step's other 4 ms are not in this function; profiles/NOTES_r6.md section 2.)"""
    import torch.distributed as dist
    if not multi():
        return
    bufs = [b for b in module.buffers() if b.is_floating_point()]
    if not bufs:
        return
    key = tuple(b.data_ptr() for b in bufs)
    plan = _BUFFER_PLANS.get(id(module))
    if plan is None or plan[0] != key:
        flat = torch.empty(sum(b.numel() for b in bufs), dtype=torch.float32, device=bufs[0].device)
        views, off = [], 0
        for b in bufs:
            views.append(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
        plan = (key, flat, views)
        if len(_BUFFER_PLANS) > 8:
            _BUFFER_PLANS.clear()
        _BUFFER_PLANS[id(module)] = plan
    _, flat, views = plan
    data = [b.data for b in bufs]
    torch._foreach_copy_(views, data)
    dist.broadcast(flat, src=src)
    torch._foreach_copy_(data, views)


class GradBuckets:
    """Data-parallel gradient averaging for fit (one process per GPU, shapes sharded over the ranks).

    All gradients end up in a few large flat fp32 buffers, in the order the backward pass produces them (decoder first, encoder
    last).  Autograd hands every parameter its own gradient tensor (`p.grad` is None when backward starts, so nothing is added
    to anything: accumulating into 298 pre-existing views cost 298 four-microsecond `add_` launches per step); when the last
    gradient of a bucket has arrived, ONE multi-tensor copy moves the bucket's gradients into its flat buffer, `p.grad` is
    re-pointed at the views and the bucket's all-reduce is issued asynchronously, so the collective of the decoder bucket
    overlaps the encoder's backward; `finish()` launches whatever is left (buckets holding parameters that got no gradient
    this step -- they contribute zeros, identically on every rank), waits, and divides by the world size.  Few, large
    messages: a ring all-reduce over xGMI is per-link bound (~153 GB/s), so 55 MB of fp32 gradients cost ~1 ms as 3 buckets
    and far more as 455 per-tensor collectives."""

    def __init__(self, params, n_buckets=3, defer=False, comm_dtype=None, groups=None):
        """groups: explicit buckets (a list of parameter lists; default: `n_buckets` of equal size in reverse registration order).  fit passes the
            groups of train_graph.parameter_stages: bucket k = the parameters whose gradients backward stage k completes.
        defer: no collective inside backward -- the hooks only count (they do not fire in a replayed HIP graph anyway).  The caller issues the
            collectives itself: reduce(k) right after the piece of the backward pass that completes bucket k (fit: after the replay of sub-graph k,
            so that the all-reduce of bucket k overlaps sub-graph k + 1 -- what DDP's hooks do in the reference's Lightning run); whatever was not
            issued that way goes out in finish().
        comm_dtype (default: environment PPS_GRAD_BUCKET_DTYPE, e.g. 'bf16'): the buckets are summed over the ranks in this type (half the xGMI
            bytes, 27.5 instead of 55 MB per step); the flat fp32 buffers the optimizer reads stay fp32."""
        import torch.distributed as dist
        self.dist = dist
        self.defer = bool(defer)                    # True: no collective inside backward (fit with the forward / backward replayed as a HIP graph)
        if comm_dtype is None:
            comm_dtype = {'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16, 'f16': torch.float16, '': None, 'f32': None}[os.environ.get('PPS_GRAD_BUCKET_DTYPE', '')]
        self.comm_dtype = comm_dtype
        self.params = [p for p in params if p.requires_grad]
        if groups is not None:
            # an empty group stays an (empty) bucket: bucket k == backward stage k for fit.StagedStep, whatever is frozen
            self.buckets = [[p for p in g if p.requires_grad] for g in groups]
            ids = [id(p) for b in self.buckets for p in b]
            if len(ids) != len(set(ids)) or set(ids) != {id(p) for p in self.params}:
                raise ValueError('GradBuckets: `groups` must partition the parameters that require a gradient')
        else:
            rev = list(reversed(self.params))                                 # roughly the order of gradient production
            total = sum(p.numel() for p in rev)
            self.buckets, cur, size = [], [], 0
            for p in rev:
                cur.append(p)
                size += p.numel()
                if size >= total / n_buckets and len(self.buckets) < n_buckets - 1:
                    self.buckets.append(cur)
                    cur, size = [], 0
            if cur:
                self.buckets.append(cur)
        self.flat, self.views, self.pending, self.handles, self.launched = [], [], [], [], []
        for bi, bucket in enumerate(self.buckets):
            flat = torch.zeros(sum(p.numel() for p in bucket), dtype=torch.float32, device=(bucket[0] if bucket else self.params[0]).device)
            off, views = 0, []
            for p in bucket:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
                p.grad = None
                p.register_post_accumulate_grad_hook(self._hook(bi))
            self.flat.append(flat)
            self.views.append(views)
        self.low = [torch.empty_like(f, dtype=comm_dtype) for f in self.flat] if comm_dtype is not None else None
        self._expect = None                         # per bucket: parameters expected to get a gradient (known after the first finish())
        self._use_expect = os.environ.get('PPS_GRAD_EXPECT', '1') != '0'
        self._static = None                         # host copy of the first step's global "got a gradient" mask
        self._static_dev = None
        self._mismatch = None                       # device flag: a later step's global mask differed from the first step's
        self._steps = 0
        self.order_log = None                       # tests: a list that receives 'reduce<k>' when bucket k's collective is issued
        self.hold = False                           # measurement (bench.py): reduce(k) does nothing, every collective goes out in finish() -- the
                                                    # step WITHOUT all-reduce / backward overlap, to price the overlap against
        self._reset()
        self._armed = False                         # hooks stay inert until the first zero(): a backward pass outside zero() ... finish() is not ours

    def _reset(self):
        # a bucket is complete when every parameter of it that is EXPECTED to get a gradient has one.  Before the first finish() that is every
        # parameter; afterwards the parameters no rank touched in the first step (POCO's encoder.cv5 / bn5 never get a gradient: the train_poco
        # fixture lists them under 'unused') are no longer waited for -- otherwise their bucket would never complete inside backward and, with
        # collectives issued in bucket order, neither would any bucket behind it
        if self._expect is None:
            self.pending = [len(b) for b in self.buckets]
        else:
            self.pending = list(self._expect)
        self.handles = []
        self.launched = [False] * len(self.buckets)
        self.reduced = [False] * len(self.buckets)  # the bucket's collective has been issued
        self.touched = set()
        self._replay = False
        self._next = 0                              # first bucket whose collective has not been issued yet
        self._armed = True

    def _hook(self, bi):
        def fn(p):
            if not self._armed:                     # not between zero() and finish(): a backward pass this bucket set is not part of
                return
            if self.launched[bi]:
                # the bucket went out without this gradient: it can only happen when a parameter that got no gradient on any rank in the first
                # step gets one now.  Failing loudly beats stepping on a gradient that missed the all-reduce.
                raise RuntimeError('GradBuckets: a gradient arrived for a parameter of bucket {} after the bucket was packed (the set of parameters '
                                   'with gradients changed between steps); set PPS_GRAD_EXPECT=0 to wait for every parameter'.format(bi))
            self.touched.add(id(p))
            self.pending[bi] -= 1
            if self.pending[bi] <= 0:
                # collectives are issued in bucket order 0, 1, 2 on EVERY rank, whatever the order in which a rank's buckets fill up (a rank
                # that misses a gradient of bucket 0 must not start with bucket 1 while the others start with bucket 0): a complete bucket
                # waits for its predecessors, finish() issues what is left, in order
                while self._next < len(self.buckets) and self.pending[self._next] <= 0:
                    self._launch(self._next)
                    self._next += 1
        return fn

    def pack_all(self):
        """Copies every bucket that has not been packed yet into its flat buffer, in bucket order, WITHOUT any collective.  Called at the end of a
        step body that is recorded into a HIP graph (defer=True): every bucket -> flat copy then is part of the graph, whichever buckets the hooks
        completed during the recorded backward (a bucket holding a parameter without gradient never completes in the hooks)."""
        for bi in range(len(self.buckets)):
            if not self.launched[bi]:
                self._launch(bi, collective=False)
        self._next = len(self.buckets)

    def _launch(self, bi, collective=True):
        _, ws = world()
        self.launched[bi] = True
        pairs = [(v, p) for v, p in zip(self.views[bi], self.buckets[bi]) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if pairs:
            torch._foreach_copy_([v for v, _ in pairs], [p.grad for _, p in pairs])
            for v, p in pairs:
                p.grad = v
        if multi() and not self.defer and collective:
            self.reduced[bi] = True
            self.handles.append(self._all_reduce(bi))

    def pack(self, bi):
        """Bucket bi -> its flat buffer, no collective (the end of a backward stage that is being recorded into a HIP graph)."""
        if not self.launched[bi]:
            self._launch(bi, collective=False)

    def reduce(self, bi):
        """Issue the (asynchronous) all-reduce of bucket bi now; finish() waits for it.  Buckets go out in index order on every rank."""
        if not multi() or self.reduced[bi] or self.hold:
            return
        assert all(self.reduced[:bi]), 'gradient buckets are all-reduced in index order'
        self.pack(bi)
        self.reduced[bi] = True
        self.handles.append(self._all_reduce(bi))
        if self.order_log is not None:
            self.order_log.append('reduce{}'.format(bi))

    def _all_reduce(self, bi):
        """Asynchronous sum of bucket bi over the ranks (in comm_dtype if set); returns something with .wait()."""
        flat = self.flat[bi]
        if flat.numel() == 0:                       # a stage without trainable parameters: nothing on the wire (the same decision on every rank)
            class _Nothing:
                @staticmethod
                def wait():
                    pass
            return _Nothing
        if self.low is None:
            return self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, async_op=True)
        low = self.low[bi]
        low.copy_(flat)
        h = self.dist.all_reduce(low, op=self.dist.ReduceOp.SUM, async_op=True)

        class _Back:
            @staticmethod
            def wait():
                h.wait()
                flat.copy_(low)
        return _Back

    def begin_replay(self, touched):
        """Host-side start of a step whose forward / backward will be REPLAYED from HIP graphs (no Python runs: neither zero() nor the hooks): the
        buckets are packed by the graphs themselves, `touched` = the parameters that received a gradient when the graphs were recorded."""
        self.handles = []
        self.reduced = [False] * len(self.buckets)
        self.replayed(touched)
        self._armed = True

    def replayed(self, touched):
        """The forward / backward of this step ran as a replayed HIP graph: none of the Python hooks fired.  `touched` = ids of the
        parameters that received a gradient when the graph was captured (their p.grad already are the bucket views)."""
        self.touched = set(touched)
        self.launched = [True] * len(self.buckets)
        self._replay = True                         # the set was fixed at capture time, identically on every rank (same graph everywhere)

    def zero(self):
        """Replaces optimizer.zero_grad(): flat buffers cleared (slots of parameters without a gradient stay zero), every p.grad None."""
        for flat, bucket in zip(self.flat, self.buckets):
            flat.zero_()
            for p in bucket:
                p.grad = None
        self._reset()

    def finish(self):
        _, ws = world()
        self._armed = False
        for bi in range(len(self.buckets)):
            if not self.launched[bi]:
                self._launch(bi)
        on = multi()
        if on:                                      # whatever the caller (defer) or the hooks have not sent yet, in index order
            for bi in range(len(self.buckets)):
                if not self.reduced[bi]:
                    self.reduced[bi] = True
                    self.handles.append(self._all_reduce(bi))
        for h in self.handles:
            h.wait()
        if on:
            for flat in self.flat:
                flat.div_(ws)
        touched = self.touched
        if on:
            # the decision "this parameter got a gradient" must be the same on every rank, or some replicas would step the parameter (averaged
            # gradient, weight decay, step count) and others skip it: one tiny MAX all-reduce of a per-parameter mask per step.  A
            # parameter touched on ANY rank keeps its averaged gradient everywhere (ranks that did not touch it contributed zeros).  The collective
            # is issued on EVERY step, replayed or not, so that a rank whose capture failed (eager) and a rank that replays still run the same
            # sequence of collectives.  Only the first step reads the mask back (one host synchronisation per run); later steps
            # compare it with the first step's mask on the device and the flag is read every CHECK_EVERY steps.
            mask = self._touched_mask(touched)
            self.dist.all_reduce(mask, op=self.dist.ReduceOp.MAX)
            if self._static is None:
                self._static = [bool(k) for k in (mask.cpu().numpy() > 0)]
                self._static_dev = mask.clone()
                self._mismatch = torch.zeros((), dtype=torch.bool, device=mask.device)
            else:
                self._mismatch |= (mask != self._static_dev).any()
                self._steps += 1
                if self._steps % self.CHECK_EVERY == 0:
                    self.check()
            touched = {id(p) for p, k in zip(self.params, self._static) if k}
            for views, bucket in zip(self.views, self.buckets):
                for v, p in zip(views, bucket):
                    if id(p) in touched and p.grad is None:
                        p.grad = v                   # touched elsewhere only: the averaged gradient sits in this rank's bucket slot
        if self._expect is None and self._use_expect:
            self._expect = [sum(1 for p in bucket if id(p) in touched) for bucket in self.buckets]
        for p in self.params:                       # like plain autograd: no gradient -> the optimizer skips the parameter
            if id(p) not in touched:                # (AdamW would otherwise still apply weight decay to it)
                p.grad = None

    CHECK_EVERY = 64

    def _touched_mask(self, touched):
        """Per-parameter int32 "got a gradient" mask on the device WITHOUT a blocking upload per step (torch.tensor(list, device=cuda) is a
        pageable copy followed by a stream synchronise, so every step waited for the replayed backward and all bucket collectives before the host
        could queue the optimizer).  The local set is the same step after step (fixed at capture for a replayed step): its device image is cached
        and only re-uploaded -- from a pinned staging buffer, asynchronously -- when the set differs from the previous step's.  The collective
        reduces a device-to-device copy of it in place."""
        key = tuple(id(p) in touched for p in self.params)
        dev = self.flat[0].device
        if getattr(self, '_mask_key', None) != key:
            host = torch.tensor([1 if k else 0 for k in key], dtype=torch.int32)
            if dev.type == 'cuda':
                if getattr(self, '_mask_pin', None) is None:
                    self._mask_pin = torch.empty(len(self.params), dtype=torch.int32).pin_memory()
                    self._mask_src = torch.empty(len(self.params), dtype=torch.int32, device=dev)
                    self._mask_work = torch.empty_like(self._mask_src)
                    self._mask_uploaded = torch.cuda.Event()
                else:
                    self._mask_uploaded.synchronize()        # the previous upload has left the staging buffer (only when the set changed again)
                self._mask_pin.copy_(host)
                self._mask_src.copy_(self._mask_pin, non_blocking=True)
                self._mask_uploaded.record()
            else:
                self._mask_src = host.to(dev)
                self._mask_work = torch.empty_like(self._mask_src)
            self._mask_key = key
        self._mask_work.copy_(self._mask_src)
        return self._mask_work

    def check(self):
        """Raises if the set of parameters with a gradient (on any rank) ever differed from the first step's (one host synchronisation)."""
        if self._mismatch is not None and bool(self._mismatch):
            raise RuntimeError('GradBuckets: the set of parameters that received a gradient changed after the first step')
