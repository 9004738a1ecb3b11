"""Data-parallel gradient exchange for the training step (one all-reduce(mean) of the gradients per
iteration; the reference uses torch DistributedDataParallel, core/trainer.py:88-104).

One process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI) -- or "gloo" on CPU for the tests.  Design for
xGMI (point-to-point links, ring collectives are per-link bound): few, LARGE flat buckets (default 128 MiB of
fp32 gradients: the whole 344 MB ViT-B gradient is 3 all-reduces, each long enough to reach link bandwidth) launched as soon
as the last gradient of a bucket has been produced, on a side stream, so the exchange of the late layers overlaps the
backward of the early ones.  Buckets are filled in REVERSE parameter order (the order the backward pass produces gradients).

    reducer = GradReducer(model.parameters())
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
    def __init__(self, params, bucket_bytes=128 << 20, process_group=None, average=True, always_bucket=False):
        self.params = [p for p in params if p.requires_grad]
        self.always_bucket = always_bucket       # pack into flat buckets even at world size 1 (tests / flat-gradient optimizers)
        self.group = process_group
        self.average = average
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.buckets = []                       # dicts: params, offsets, numel, flat (lazy), pending, work
        cur, cur_bytes = [], 0
        for p in reversed(self.params):         # backward order
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._slot = {}                          # id(param) -> (bucket, index in bucket); tensors must not be compared with ==
        for b in self.buckets:
            for i, p in enumerate(b['params']):
                self._slot[id(p)] = (b, i)
        self._stream = None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _close(self, ps):
        offs, n = [], 0
        for p in ps:
            offs.append(n)
            n += p.numel()
        self.buckets.append(dict(params=ps, offsets=offs, numel=n, flat=None, pending=len(ps), work=None, event=None))

    def _on_grad(self, p):
        if self.world == 1 and not self.always_bucket:
            return                               # single process: nothing to exchange, the gradients stay where autograd put them
        b, i = self._slot[id(p)]
        b['pending'] -= 1
        if b['pending'] == 0:
            # the bucket's last gradient has arrived: ONE multi-tensor copy packs all of them (a copy per hook was ~225 small launches
            # per step: 1.3 ms of the batch-64 W-HMR step), then the exchange starts
            flat = b['flat'] = torch.empty(b['numel'], dtype=torch.float32, device=p.grad.device)
            views = [flat[off:off + q.numel()].view_as(q) for q, off in zip(b['params'], b['offsets'])]
            torch._foreach_copy_(views, [q.grad for q in b['params']])
            self._launch(b)

    def _launch(self, b):
        if self.world == 1:
            return
        flat = b['flat']
        if flat.is_cuda:
            # the exchange runs on a side stream so that the rest of the backward keeps the compute stream busy
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            self._stream.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._stream):
                b['work'] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b['work'] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Wait for every bucket, apply the 1/world mean, and make ``p.grad`` views of the reduced buckets."""
        if self.world == 1 and not self.always_bucket:
            return
        for b in self.buckets:
            if b['pending'] != 0:
                missing = [tuple(p.shape) for p in b['params'] if p.grad is None]
                raise RuntimeError('backward left %d gradient(s) of a bucket unset (shapes %s)' % (b['pending'], missing[:4]))
            if b['work'] is not None:
                b['work'].wait()
                if b['flat'].is_cuda:
                    torch.cuda.current_stream(b['flat'].device).wait_stream(self._stream)
            if self.average and self.world > 1:
                b['flat'].div_(self.world)
            for p, off in zip(b['params'], b['offsets']):
                p.grad = b['flat'][off:off + p.numel()].view_as(p)
            b['pending'], b['work'] = len(b['params']), None
        # the flat buffers now back the gradients: allocate fresh ones on the next step's first hook
        for b in self.buckets:
            b['flat'] = None

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
    flat = torch.cat([b.detach().reshape(-1).float() for _, b in named])
    dist.broadcast(flat, src=src, group=process_group)
    off = 0
    with torch.no_grad():
        for _, b in named:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
    return len(named)
