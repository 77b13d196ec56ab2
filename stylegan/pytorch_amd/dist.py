"""Data parallelism for the G+D step: one process per MI355X, gradients summed with RCCL over xGMI.

The reference is single-device (README.md:30 / models/GAN.py:509-510 are TODOs); the parity target is the
single-device result at the GLOBAL batch:

* the batch is sharded so every rank holds whole minibatch-stddev groups (``stddev_preserving_shard``): the
  reference groups sample i with i+M, i+2M, i+3M (M = B/4, models/CustomLayers.py:296-297);
* gradients are all-reduced with SUM.  The softplus loss terms are batch MEANS (models/Losses.py:218,229) and are
  pre-scaled by 1/world_size (Losses.LogisticGAN.mean_scale); the R1 term is a batch SUM (:210) and is not;
* the generator all-reduce happens BEFORE the global-norm clip (models/GAN.py:651), so the norm is the global one;
* ``Truncation.update`` uses global sample 0 (models/GAN.py:278) = rank 0's local sample 0: the buffer is broadcast.

Collective choice for xGMI (7 point-to-point links per GPU): few large flat buckets (default 32 MiB; the D and G
gradient sets are ~92 / ~105 MB at 1024x1024).  From the second iteration at a depth on the gradients LIVE in those
buckets (``GradBuckets``: every ``.grad`` is a view into a flat buffer), so the all-reduce runs on the buffers in place;
the first iteration (active set still unknown) concatenates and copies back.
"""
import torch
import torch.distributed as dist


def stddev_preserving_shard(global_batch: int, world_size: int, rank: int, group_size: int = 4):
    """Indices of the global batch owned by ``rank`` such that local StddevLayer groups == global groups."""
    g = min(group_size, global_batch)
    assert global_batch % g == 0, "batch must be divisible by the stddev group size"
    m = global_batch // g
    assert m % world_size == 0, f"need batch % ({g}*world_size) == 0, got batch {global_batch}, world {world_size}"
    per = m // world_size
    return [gi * m + mi for gi in range(g) for mi in range(rank * per, (rank + 1) * per)]


def bucketize(sizes, bucket_elems):
    """Greedy partition of consecutive tensors into buckets of at most ``bucket_elems`` elements (>= 1 tensor each)."""
    buckets, cur, cur_n = [], [], 0
    for i, n in enumerate(sizes):
        if cur and cur_n + n > bucket_elems:
            buckets.append(cur); cur, cur_n = [], 0
        cur.append(i); cur_n += n
    if cur:
        buckets.append(cur)
    return buckets


class GradBuckets:
    """Flat gradient storage for a FIXED set of parameters (the active set of one network at one progressive depth): a few
    large fp32 buffers; every parameter's ``.grad`` is a view into one of them, so the all-reduce runs on the buffers
    themselves -- no concatenation before, no copy back after (two passes over ~100 MB per network and step otherwise).
    ``attach()`` zero-fills the buffers and installs the views; the backward then accumulates into them (the convolution
    weight-gradient kernels and autograd's AccumulateGrad both add in place into an existing ``.grad``)."""

    ALIGN = 64                                               # elements: every view starts on a 256-byte boundary

    def __init__(self, params, bucket_elems):
        self.params = list(params)
        assert self.params, "GradBuckets needs at least one parameter"
        dtype = self.params[0].dtype                        # fp32 in the product (parameters and their gradients are fp32)
        assert all(p.dtype == dtype for p in self.params)
        dev = self.params[0].device
        pad = lambda n: (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.buckets = []
        for idx in bucketize([pad(p.numel()) for p in self.params], bucket_elems):
            chunk = [self.params[i] for i in idx]
            flat = torch.zeros(sum(pad(p.numel()) for p in chunk), dtype=dtype, device=dev)
            views, off = [], 0
            for p in chunk:
                views.append((p, flat[off:off + p.numel()].view(p.shape)))
                off += pad(p.numel())
            self.buckets.append((flat, views))

    def matches(self, params):
        """True if ``params`` (the parameters that received a gradient) are exactly this layout's parameters (in any order: a
        layout built by ``BucketScheduler`` is in gradient-ready order, not in parameter order)."""
        params = list(params)
        return len(params) == len(self.params) and {id(p) for p in params} == {id(p) for p in self.params}

    def attach(self):
        for flat, views in self.buckets:
            flat.zero_()
            for p, v in views:
                p.grad = v

    def attached(self):
        return all(p.grad is v for _, views in self.buckets for p, v in views)


class BucketScheduler:
    """Bucket-level overlap of the gradient all-reduce with the backward that produces the gradients (the reference's
    multi-GPU TODO, models/GAN.py:509-510).

    The backward of a half-iteration writes every active parameter's gradient a FIXED number of times in a FIXED order (one
    graph per (network, depth): the discriminator's parameters get up to three contributions -- fake pass, real pass, R1 double
    backward).  ``record`` mode (first iteration at a depth) counts the contributions per parameter and notes the order in which
    the parameters become FINAL; ``GradBuckets`` is then laid out in that order, so a bucket fills up early, and from the next
    iteration on ``note(p)`` -- called by the autograd post-accumulate hook of ``p`` or, for convolution parameters whose
    gradient is accumulated inside the weight-gradient kernel, by functional.ConvFn.backward -- fires the bucket's all-reduce the
    moment its last parameter is final: on the collective side stream, after the stream(s) that wrote the gradients, while the
    backward carries on with the higher-resolution layers.  ``finish`` fires what is left and joins.  A parameter that is
    written MORE often than recorded would be reduced too early: ``finish`` checks every count and raises (the step's
    gradients are then unusable; ``StyleGAN._reduce`` drops the schedule and the layout so that the next backward records again).

    Cross-rank agreement (round 4).  The order in which THIS rank saw gradients become final is not a safe basis for a layout:
    autograd sums the contributions of a parameter that is used several times in an order that depends on thread-local
    sequence numbers, so two ranks can record different last-write orders -- and buckets with different contents or sizes mean
    mismatched collectives (a hang or silent corruption under RCCL).  So (i) ``layout`` takes RANK 0's recorded order: rank 0
    broadcasts (canonical parameter index, contribution count) pairs, every rank lays its buckets out from that list and checks
    its own recorded set and counts against it (a mismatch raises on ALL ranks together: the verdict is all-reduced); (ii) buckets
    are FIRED IN BUCKET ORDER -- a bucket that completes before an earlier one waits for it (as torch DDP does) -- so every rank
    issues the same sequence of collectives whatever its local completion order; (iii) collectives are issued under the
    scheduler's lock, during the backward only from autograd's hook calls and after it only from the caller's thread (the
    backward call blocks that thread), so the per-process issue order is a total order."""

    def __init__(self, group, params=None):
        self.group = group
        self.lock = __import__("threading").Lock()
        self.recording = params is None
        self.count, self.order = {}, []                       # recording: id(p) -> contributions, ids in order of LAST write
        self.gb = None
        self._params = {}
        self.order_source = "local"                           # "rank0" once the layout came from rank 0's broadcast

    # ---- recording (first iteration at a depth)
    def note(self, p):
        with self.lock:
            k = id(p)
            if self.recording:
                self._params[k] = p
                self.count[k] = self.count.get(k, 0) + 1
                if k in self.order:
                    self.order.remove(k)
                self.order.append(k)
                return
            c = self.seen.get(k)
            if c is None:
                self.unknown += 1                              # a parameter outside the recorded set received a gradient
                return
            self.seen[k] = c + 1
            b = self.bucket_of[k]
            if p.is_cuda:
                # the stream this contribution was enqueued on (AccumulateGrad runs on the stream of the parameter's forward
                # use: with the fake branch of the D step on the auxiliary stream a bucket can hold gradients written on two
                # streams); the bucket's all-reduce is ordered behind every one of them
                st = torch.cuda.current_stream(p.device)
                self.streams[b][st.cuda_stream] = st
            if c + 1 == self.count[k]:
                self.left[b] -= 1
                if self.left[b] == 0:
                    self.ready[b] = True
                    self._fire_ready()

    def _fire_ready(self):
        """Fire, in BUCKET ORDER, every complete bucket whose predecessors have all been fired (lock held)."""
        while self.next_fire < len(self.ready) and self.ready[self.next_fire]:
            self._fire(self.next_fire)
            self.next_fire += 1

    def _agree_on_order(self, ids, canonical):
        """Rank 0's gradient-ready order for everybody.  ``ids``: this rank's recorded order; ``canonical``: the parameters in an
        order every rank shares (``net.parameters()``).  Returns the ids in rank 0's order; raises on ALL ranks if any rank's
        recorded set or contribution counts differ from rank 0's."""
        g = self.group
        index_of = {id(p): i for i, p in enumerate(canonical)}
        if any(k not in index_of for k in ids):
            raise RuntimeError("BucketScheduler.layout: a recorded parameter is not in the canonical parameter list")
        mine = [(index_of[k], self.count[k]) for k in ids]
        dev = canonical[0].device if (canonical and dist.get_backend(g.group) == "nccl") else torch.device("cpu")
        n = torch.tensor([len(mine)], dtype=torch.int64, device=dev)
        dist.broadcast(n, src=0, group=g.group)
        table = torch.tensor(mine if g.rank == 0 else [[0, 0]] * int(n.item()), dtype=torch.int64, device=dev).reshape(-1, 2)
        dist.broadcast(table, src=0, group=g.group)
        theirs = [(int(i), int(c)) for i, c in table.cpu().tolist()]
        ok = sorted(mine) == sorted(theirs)
        verdict = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=g.group)
        if int(verdict.item()) != 1:
            raise RuntimeError("BucketScheduler.layout: the ranks recorded different gradient sets or contribution counts "
                               f"(rank {g.rank}: {'matches' if ok else 'differs from'} rank 0); no bucket layout was built")
        self.order_source = "rank0"
        return [id(canonical[i]) for i, _ in theirs]

    def layout(self, bucket_elems, only=None, canonical=None):
        """After a recording backward: GradBuckets of the recorded parameters (``only``: restrict to these, e.g. the ones that
        really hold a gradient) in gradient-ready order, and this object switched to firing mode for the next backward.
        ``canonical`` (data parallel: REQUIRED for more than one rank): the network's parameters in an order all ranks share;
        the layout then follows rank 0's gradient-ready order (class docstring)."""
        keep = None if only is None else {id(p) for p in only}
        ids = [k for k in self.order if keep is None or k in keep]
        g = self.group
        multi = g is not None and (g.world_size > 1 or g.force_collectives)
        if multi:
            if canonical is None:
                raise RuntimeError("BucketScheduler.layout: more than one rank needs the canonical parameter list")
            ids = self._agree_on_order(ids, list(canonical))      # (also with an empty list: the collectives must match)
        if not ids:
            return None
        self.gb = GradBuckets([self._params[k] for k in ids], bucket_elems)
        self.bucket_of = {}
        for bi, (_, views) in enumerate(self.gb.buckets):
            for p, _ in views:
                self.bucket_of[id(p)] = bi
        self.count = {k: self.count[k] for k in ids}
        self.recording = False
        self._params = {}
        return self.gb

    # ---- firing (later iterations)
    def begin(self, param_stream=None):
        """Before the backward.  ``param_stream``: the side stream the convolution weight gradients are written on (the bucket's
        all-reduce must be ordered after it as well as after the stream of the backward chain)."""
        self.seen = {k: 0 for k in self.count}
        self.left = [len(views) for _, views in self.gb.buckets]
        self.fired = [False] * len(self.gb.buckets)
        self.ready = [False] * len(self.gb.buckets)
        self.next_fire = 0
        self.streams = [dict() for _ in self.gb.buckets]      # per bucket: raw handle -> stream its gradients were written on
        self.unknown = 0
        self.handles = []
        self.param_stream = param_stream
        self.fired_early = 0

    def _fire(self, b):
        flat = self.gb.buckets[b][0]
        self.fired[b] = True
        g = self.group
        if g.world_size == 1 and not g.force_collectives:
            return
        if flat.is_cuda:
            side = g._side_stream(flat.device)
            cur = torch.cuda.current_stream(flat.device)
            side.wait_stream(cur)
            if self.param_stream is not None:
                side.wait_stream(self.param_stream)
            for raw, st in self.streams[b].items():            # every stream a gradient of this bucket was written on
                if raw != cur.cuda_stream and (self.param_stream is None or raw != self.param_stream.cuda_stream):
                    side.wait_stream(st)
            with torch.cuda.stream(side):
                g._all_reduce(flat)
            flat.record_stream(side)
        else:
            self.handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=g.group, async_op=True))

    def finish(self):
        """After the backward: all-reduce the buckets that did not complete early (none, if the recording still holds), join
        the collective stream / the pending handles, and verify that every parameter was written exactly as often as recorded."""
        with self.lock:
            self.fired_early = sum(self.fired)
            bad = [k for k, c in self.seen.items() if c != self.count[k]]
            late = [b for b, f in enumerate(self.fired) if not f]
            if self.unknown or any(self.seen[k] > self.count[k] for k in bad):
                raise RuntimeError("BucketScheduler: the backward wrote gradients that the recorded schedule does not know "
                                   f"({self.unknown} unknown parameters, {len(bad)} with another count): a bucket may have been reduced early")
            for b in late:                                     # parameters that got FEWER writes this time (or none): reduce
                self.ready[b] = True                           # now -- still in bucket order
            self._fire_ready()
        for h in self.handles:
            h.wait()
        self.handles = []
        flat0 = self.gb.buckets[0][0]
        if flat0.is_cuda and (self.group.world_size > 1 or self.group.force_collectives):
            torch.cuda.current_stream(flat0.device).wait_stream(self.group._side_stream(flat0.device))


_ACTIVE_SCHEDULER = None            # the scheduler of the backward in flight (one process drives one GPU)


def set_active_scheduler(s):
    global _ACTIVE_SCHEDULER
    _ACTIVE_SCHEDULER = s


def note_grad_write(p):
    """A gradient contribution to ``p`` has been enqueued (on the current stream, or on the weight-gradient side stream)."""
    s = _ACTIVE_SCHEDULER
    if s is not None:
        s.note(p)


def install_grad_hooks(params):
    """Post-accumulate hooks (autograd's AccumulateGrad) for parameters whose gradient travels through autograd; convolution
    parameters accumulated in-kernel are noted by functional.ConvFn.backward.  Idempotent."""
    for p in params:
        if getattr(p, "_sgx_grad_hook", None) is None and p.requires_grad:
            p._sgx_grad_hook = p.register_post_accumulate_grad_hook(note_grad_write)


class _GlobalMeanFn(torch.autograd.Function):
    """mean over the GLOBAL batch of a per-rank tensor holding B/N samples: m = (1/N) sum_k mean(x_k).

    With L = sum_r L_r (every rank's loss already carries 1/N, Losses.GANLoss.mean_scale) and every L_r a function of m,
    dL/dmean(x_k) = (1/N) sum_r dL_r/dm: the backward is the same all-reduce of the incoming gradient."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group, ctx.shape = group, x.shape
        return group._mean_over_ranks(x.mean())

    @staticmethod
    def backward(ctx, g):
        gm = ctx.group._mean_over_ranks(g)
        n = 1
        for d in ctx.shape:
            n *= d
        return (gm / n).expand(ctx.shape), None


class DataParallelGroup:
    """Bucketed gradient all-reduce(SUM) + buffer broadcast over a torch.distributed process group
    (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests)."""

    def __init__(self, group=None, bucket_mb: float = 32.0, force_collectives: bool = False, overlap_buckets: bool = True):
        assert dist.is_initialized()
        self.group = group
        self.force_collectives = force_collectives          # tests: issue the collectives even in a group of one rank
        self.overlap_buckets = overlap_buckets               # BucketScheduler: all-reduce a bucket as soon as its gradients are final
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self._side = None
        # "gloo" cannot reduce device memory: GPU tensors are staged through the host (the 2-process-on-one-GPU parity test
        # of StyleGAN(data_parallel=...) runs this way; production is backend "nccl" = RCCL, device to device over xGMI)
        self._stage_through_host = dist.get_backend(group) == "gloo"

    def _all_reduce(self, t):
        if self._stage_through_host and t.is_cuda:
            h = t.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    @torch.no_grad()
    def all_reduce_buckets(self, gb: GradBuckets):
        """Sum the flat gradient buffers of ``gb`` over ranks, in place, on the side stream (joined before returning to the
        caller's stream order)."""
        if self.world_size == 1 and not self.force_collectives:
            return
        dev = gb.buckets[0][0].device
        side = self._side_stream(dev)
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(dev))
        for flat, _ in gb.buckets:
            if side is not None:
                with torch.cuda.stream(side):
                    self._all_reduce(flat)
                flat.record_stream(side)
            else:
                self._all_reduce(flat)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)

    @torch.no_grad()
    def all_reduce_scalar(self, t):
        """Sum of a (loss) scalar over ranks, as a new tensor on the current stream."""
        out = t.detach().clone()
        if self.world_size > 1 or self.force_collectives:
            self._all_reduce(out)
        return out

    def host_max(self, values):
        """Element-wise MAX of a list of host floats over the ranks (a HOST decision every rank must take identically: launch mode
        of a depth, "did every rank's capture succeed").  Synchronous; called a handful of times per depth, never per iteration."""
        vals = [float(v) for v in values]
        if self.world_size == 1 and not self.force_collectives:
            return vals
        if dist.get_backend(self.group) == "gloo" or not torch.cuda.is_available():
            t = torch.tensor(vals, dtype=torch.float64)
        else:
            t = torch.tensor(vals, dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return [float(v) for v in t.cpu()]

    def all_ok(self, ok: bool) -> bool:
        """True iff ``ok`` on EVERY rank."""
        return self.host_max([0.0 if ok else 1.0])[0] == 0.0

    def _mean_over_ranks(self, t):
        out = t.detach().clone()
        if self.world_size > 1 or self.force_collectives:
            self._all_reduce(out)
            out = out / self.world_size
        return out

    def global_mean(self, x):
        """Differentiable mean of ``x`` (this rank's B/N samples) over the global batch -- what ``torch.mean`` of the
        single-device batch is to the reference's relativistic loss (models/Losses.py:166-167,183-184).  Equal shard sizes
        (stddev_preserving_shard) make it the mean of the rank means."""
        return _GlobalMeanFn.apply(x, self)

    def _side_stream(self, device):
        if device.type != "cuda":
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    @torch.no_grad()
    def all_reduce_grads(self, params):
        """Sum ``p.grad`` over ranks for every parameter that has one (the active set is identical on all ranks:
        the progressive depth is global)."""
        grads = [p.grad for p in params if p.grad is not None]
        if not grads or (self.world_size == 1 and not self.force_collectives):
            return
        dev = grads[0].device
        side = self._side_stream(dev)
        flats = []
        for idx in bucketize([g.numel() for g in grads], self.bucket_elems):
            chunk = [grads[i] for i in idx]
            flat = torch.cat([g.reshape(-1) for g in chunk])
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    self._all_reduce(flat)
                flat.record_stream(side)
            else:
                self._all_reduce(flat)
            flats.append((flat, chunk))
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
        for flat, chunk in flats:                             # one multi-tensor copy per bucket, not one launch per parameter
            views, off = [], 0
            for g in chunk:
                views.append(flat[off:off + g.numel()].view_as(g)); off += g.numel()
            torch._foreach_copy_(chunk, views)

    @torch.no_grad()
    def broadcast(self, tensor, src: int = 0):
        if self.world_size > 1 or self.force_collectives:
            if self._stage_through_host and tensor.is_cuda:
                h = tensor.detach().cpu()
                dist.broadcast(h, src=src, group=self.group)
                tensor.copy_(h)
            else:
                dist.broadcast(tensor, src=src, group=self.group)
