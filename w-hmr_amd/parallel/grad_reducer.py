"""Data-parallel gradient exchange for the training step (one all-reduce(mean) of the gradients per
iteration; the reference uses torch DistributedDataParallel with find_unused_parameters=True, core/trainer.py:84-104).

One process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI) -- or "gloo" on CPU for the tests.  Design for
xGMI (point-to-point links, ring collectives are per-link bound): few, LARGE flat buckets (default 128 MiB of
fp32 gradients: the whole 344 MB ViT-B gradient is 3 all-reduces, each long enough to reach link bandwidth) launched as soon
as the last gradient of a bucket has been produced, on a side stream, so the exchange of the late layers overlaps the
backward of the early ones.  Buckets are filled in REVERSE parameter order (the order the backward pass produces gradients);
``groups`` (one id per parameter) additionally closes a bucket wherever the id changes -- W-HMR passes the backbone / heads split,
because the whole ViT backward is ONE autograd node: without the split the bucket that holds the last head parameters would also
wait for every ViT gradient and nothing could overlap.

Unused parameters (W-HMR: ``global_orient.*``, ``dp_head.*`` without AUX supervision, a frozen-by-loss ``cam_model`` -- the reason the
reference asks DDP for find_unused_parameters): the graph is assumed STATIC across ranks and steps, like DDP's ``static_graph``.  The
first ``finish()`` all-reduces a one-int-per-parameter "received a gradient" mask (and raises if the ranks disagree); parameters
nobody used are dropped from the buckets for good (their ``.grad`` stays None on every rank, as with DDP, and ``self.skipped`` lists
them); a used parameter whose gradient is missing in a later step travels as zeros.  A dropped parameter that later does receive a
gradient raises, and so does a second backward pass before ``finish()``.

    reducer = GradReducer(model.parameters())
    reducer.attach(model.feature_extractor.backbone)     # optional: the ViT node publishes its gradients block by block (see publish())
    loss.backward()            # hooks copy finished gradients into their bucket and launch full buckets
    reducer.finish()           # wait, divide by world size, gradients point into the reduced buckets
    optimizer.step()
"""
import torch
import torch.distributed as dist


def shard_batch(global_batch, world_size, rank):
    """Contiguous per-rank slice [lo, hi) of a global batch (remainder spread over the first ranks)."""
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradReducer:
    def __init__(self, params, bucket_bytes=128 << 20, process_group=None, average=True, always_bucket=False, groups=None):
        params = list(params)
        if groups is not None:
            groups = [g for p, g in zip(params, groups) if p.requires_grad]
        self.params = [p for p in params if p.requires_grad]
        self.groups = groups
        self.bucket_bytes = bucket_bytes
        self.always_bucket = always_bucket       # pack into flat buckets even at world size 1 (tests / flat-gradient optimizers)
        self.group = process_group
        self.average = average
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.skipped = []                        # parameters no rank ever produced a gradient for (found at the first finish())
        self._skipped_ids = set()
        self._resolved = False                   # the used/unused census has run
        self._fired = set()                      # id(param) of the gradients that arrived since the last finish()
        self._published = set()                  # ... of those, the ones delivered through publish() (their accumulate hook is ignored)
        self._stream = None
        # bookkeeping for a self-explaining N > 1 benchmark line (bench.py): what was exchanged, and -- with `timing` on -- how long finish() kept
        # the compute stream waiting for the exchange stream (the EXPOSED part of the all-reduce; the rest ran under the backward)
        self.timing = False
        self.stats = {'finishes': 0, 'collectives': 0, 'bytes_exchanged': 0}
        self._wait_marks = []
        self._build(self.params, self.groups)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    # ---- bucket layout
    def _build(self, params, groups):
        self.buckets = []                        # dicts: params, offsets, numel, flat (lazy), pending, work
        cur, cur_bytes, cur_g = [], 0, None
        order = list(range(len(params)))[::-1]   # backward order
        for i in order:
            p, g = params[i], (groups[i] if groups is not None else None)
            nbytes = p.numel() * 4
            if cur and (cur_bytes + nbytes > self.bucket_bytes or g != cur_g):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            cur_g = g
        if cur:
            self._close(cur)
        self._next = 0                           # index of the next bucket whose collective may start (_launch_ready)
        self._slot = {}                          # id(param) -> (bucket, index in bucket); tensors must not be compared with ==
        for b in self.buckets:
            for i, p in enumerate(b['params']):
                self._slot[id(p)] = (b, i)

    def _close(self, ps):
        offs, n = [], 0
        for p in ps:
            offs.append(n)
            n += p.numel()
        self.buckets.append(dict(params=ps, offsets=offs, numel=n, flat=None, pending=len(ps), work=None, streams=set()))

    # ---- backward side
    def _active(self):
        return self.world > 1 or self.always_bucket

    def _on_grad(self, p, published=False):
        if not self._active():
            return                               # single process: nothing to exchange, the gradients stay where autograd put them
        if not published and id(p) in self._published:
            return                               # the engine still visits the accumulate node of a parameter whose node returned None
        if id(p) in self._skipped_ids:
            raise RuntimeError('GradReducer: a parameter of shape %s received a gradient after it was classified as unused at the first '
                               'finish(); the set of used parameters must be static (rebuild the reducer)' % (tuple(p.shape),))
        if id(p) in self._fired:
            raise RuntimeError('GradReducer: a second backward pass reached a parameter before finish() was called; call finish() once '
                               'per backward (gradient accumulation over several backward passes is not supported)')
        self._fired.add(id(p))
        b, i = self._slot[id(p)]
        if p.grad is not None and p.grad.is_cuda:
            # The gradient was produced on WHATEVER stream this hook / publish runs on (the heavy chain of the W-HMR step back-propagates on a
            # side stream and autograd runs a node's backward -- and its AccumulateGrad -- on the stream of its forward).  Remember WHICH: the
            # pack below orders itself behind every stream that fed its bucket, not only the one the last gradient arrived on (a
            # smaller bucket or another parameter order would otherwise pack half-written gradients with no error).  One event per (bucket,
            # stream) at pack time -- an event per parameter cost ~230 host-side records per step (one-rank RCCL smoke: +3.8 ms).
            b['streams'].add(torch.cuda.current_stream(p.grad.device))
        b['pending'] -= 1
        if b['pending'] == 0:
            self._pack(b)
            self._launch_ready()

    def publish(self, p, grad):
        """Early delivery from INSIDE an autograd node: the W-HMR backbone is one node (ViTFn), so the hooks above would see all of its ~150
        gradients at once, after its whole backward.  ``vit_backward`` hands each block's gradients over as soon as they are enqueued
        (``module.grad_sink = reducer.publish``): ``p.grad`` is set here, the bucket bookkeeping runs, a full bucket starts its all-reduce
        while the earlier blocks are still being back-propagated -- and the node returns None for that parameter.  Returns False when there
        is nothing to exchange (single process): the caller then lets autograd deliver the gradient as usual."""
        if not self._active() or id(p) not in self._slot:
            return False
        if p.grad is not None and id(p) not in self._fired:
            raise RuntimeError('GradReducer.publish: the parameter already holds a gradient (accumulation over several backward passes is not supported)')
        p.grad = grad
        self._published.add(id(p))
        self._on_grad(p, published=True)
        return True

    def attach(self, module):
        """route the early deliveries of a module whose backward is hand-driven (whmr_amd.train.vit_autograd) into this reducer"""
        module.grad_sink = self.publish
        return self

    def _launch_ready(self):
        """Collectives start in BUCKET ORDER on every rank: a packed bucket waits until all buckets before it have been launched.  With the same
        parameters used on every rank the buckets also fill in this order and nothing waits; when a rank does NOT produce some gradient in a step
        (its bucket is only completed, with zeros, by finish()) the other ranks' later buckets must not overtake it -- the ranks would issue
        their all-reduces in different orders (gloo aborts on the size mismatch, RCCL would pair unrelated buffers or hang)."""
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b['flat'] is None and not b.get('void'):
                return
            if not b.get('void'):
                self._launch(b)
            self._next += 1

    def _pack(self, b):
        # the bucket's last gradient has arrived: ONE multi-tensor copy packs all of them (a copy per hook was ~225 small launches
        # per step: 1.3 ms of the batch-64 W-HMR step), then the exchange may start.  Locally missing gradients travel as zeros.
        have = [(q, off) for q, off in zip(b['params'], b['offsets']) if q.grad is not None]
        dev = have[0][0].grad.device if have else b['params'][0].device
        full = len(have) == len(b['params'])
        if dev.type == 'cuda':
            cur = torch.cuda.current_stream(dev)
            for st in b['streams']:                  # every gradient of the bucket is complete before the pack reads it, whichever stream made it:
                if st != cur:                        # an event recorded NOW at the tail of that stream lies behind the kernels that wrote them
                    ev = torch.cuda.Event()
                    ev.record(st)
                    cur.wait_event(ev)
        b['streams'] = set()
        flat = b['flat'] = (torch.empty if full else torch.zeros)(b['numel'], dtype=torch.float32, device=dev)
        if have:
            views = [flat[off:off + q.numel()].view_as(q) for q, off in have]
            torch._foreach_copy_(views, [q.grad for q, _ in have])
        # The launch may happen LATER and from ANOTHER stream (a packed bucket waits in _launch_ready for the buckets before it; finish() launches
        # stragglers): the exchange stream orders itself behind THIS event -- the tail of the packing stream -- not behind whatever stream is
        # current at launch time (it could otherwise read a half-packed buffer).
        b['packed'] = None
        if dev.type == 'cuda':
            b['packed'] = torch.cuda.Event()
            b['packed'].record(torch.cuda.current_stream(dev))

    def _launch(self, b):
        if self.world == 1 and not (self.always_bucket and dist.is_initialized()):
            return                               # (always_bucket on an initialised one-rank group still runs the collective: the hardware smoke of
                                                 #  the exchange-stream choreography on a 1-GPU box, bench.py --always-bucket)
        flat = b['flat']
        self.stats['collectives'] += 1
        self.stats['bytes_exchanged'] += flat.numel() * 4
        if flat.is_cuda:
            # the exchange runs on a side stream so that the rest of the backward keeps the compute stream busy
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            if b.get('packed') is not None:
                self._stream.wait_event(b['packed'])             # the pack (fill + multi-tensor copy), on whichever stream it ran
            else:
                self._stream.wait_stream(torch.cuda.current_stream(flat.device))
            flat.record_stream(self._stream)                     # allocated on the packing stream, read / written by the exchange stream
            with torch.cuda.stream(self._stream):
                b['work'] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b['work'] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    # ---- unused-parameter census (first finish() only)
    def _census(self):
        """Which parameters received a gradient?  The answer must be the same on every rank (static graph): one small all-reduce
        checks that, and the parameters nobody used leave the buckets for good once this step's exchange is complete."""
        used = torch.tensor([1 if id(p) in self._fired else 0 for p in self.params], dtype=torch.int32)
        if self.world > 1:
            dev = next((p.device for p in self.params), torch.device('cpu'))
            both = torch.cat([used, -used]).to(dev)              # MAX of (u, -u) = (max u, -min u)
            dist.all_reduce(both, op=dist.ReduceOp.MAX, group=self.group)
            both = both.cpu()
            n = used.numel()
            if not torch.equal(both[:n], -both[n:]):
                raise RuntimeError('GradReducer: the ranks disagree on which parameters received a gradient (%d differ); the exchange needs a '
                                   'static graph (same used parameters on every rank)' % int((both[:n] != -both[n:]).sum()))
        self._resolved = True
        return used

    def _wait(self, b):
        if b['work'] is not None:
            b['work'].wait()
            if b['flat'].is_cuda:
                torch.cuda.current_stream(b['flat'].device).wait_stream(self._stream)
            b['work'] = None

    def finish(self):
        """Wait for every bucket, apply the 1/world mean, and make ``p.grad`` views of the reduced buckets."""
        if not self._active():
            return
        used = None
        if not self._resolved:
            used = self._census()
            for i, p in enumerate(self.params):
                if not used[i]:
                    self.skipped.append(p)
                    self._skipped_ids.add(id(p))
        self.stats['finishes'] += 1
        mark = self._mark() if self.timing else None
        for b in self.buckets:
            if b['flat'] is None:
                if all(id(p) in self._skipped_ids for p in b['params']):
                    b['void'] = True                             # nothing but unused parameters: no exchange at all
                else:
                    self._pack(b)                                # stragglers: missing gradients travel as zeros (same on every rank)
            self._launch_ready()
            if b.get('void'):
                continue
            self._wait(b)
            if self.average and self.world > 1:
                b['flat'].div_(self.world)
            for p, off in zip(b['params'], b['offsets']):
                if id(p) not in self._skipped_ids:               # unused everywhere: .grad stays None, as under DDP
                    p.grad = b['flat'][off:off + p.numel()].view_as(p)
            b['pending'], b['work'] = len(b['params']), None
        if mark is not None:
            self._wait_marks.append((mark, self._mark()))
        # the flat buffers now back the gradients: allocate fresh ones on the next step's first hook
        for b in self.buckets:
            b['flat'] = None
            b['void'] = False
        self._next = 0
        self._fired.clear()
        self._published.clear()
        if used is not None and self.skipped:                    # from the next step on the buckets hold used parameters only
            keep = [i for i in range(len(self.params)) if used[i]]
            self.groups = [self.groups[i] for i in keep] if self.groups is not None else None
            self.params = [self.params[i] for i in keep]
            self._build(self.params, self.groups)

    def _mark(self):
        dev = next((p.device for p in self.params), torch.device('cpu'))
        if dev.type == 'cuda':
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(dev))
            return ev
        import time
        return time.perf_counter()

    def exposed_wait_ms(self):
        """Per finish() since `timing` was switched on: milliseconds the compute stream (CPU: the host) spent inside finish() -- straggler packs,
        the waits for the exchange stream, the 1/world scaling -- i.e. the part of the exchange the backward did NOT hide.  Synchronises."""
        out = []
        for a, b in self._wait_marks:
            if isinstance(a, float):
                out.append((b - a) * 1e3)
            else:
                b.synchronize()
                out.append(a.elapsed_time(b))
        self._wait_marks = []
        return out

    def remove(self):
        for h in self._hooks:
            h.remove()


def broadcast_buffers(module, src=0, process_group=None):
    """DistributedDataParallel's ``broadcast_buffers=True`` (the reference wraps the model in DDP, core/trainer.py:88-104): every floating-point
    buffer -- here the BatchNorm running statistics, which each rank updates from its own batch -- is overwritten with rank ``src``'s copy.
    All buffers travel as ONE flat fp32 broadcast (a few KB for W-HMR), so calling it once per step costs one small collective."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return 0
    named = [(n, b) for n, b in module.named_buffers() if b.is_floating_point() and 'running_' in n]
    if not named:
        return 0
    bufs = [b.detach() for _, b in named]
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    dist.broadcast(flat, src=src, group=process_group)
    views, off = [], 0
    for b in bufs:
        views.append(flat[off:off + b.numel()].view_as(b))
        off += b.numel()
    with torch.no_grad():
        torch._foreach_copy_(bufs, views)            # one multi-tensor launch back (was one copy per buffer: 114 small launches per step)
    return len(named)
