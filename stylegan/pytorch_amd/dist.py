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
gradient sets are ~92 / ~105 MB at 1024x1024) issued on a side stream as soon as they are packed.
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


class DataParallelGroup:
    """Bucketed gradient all-reduce(SUM) + buffer broadcast over a torch.distributed process group
    (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests)."""

    def __init__(self, group=None, bucket_mb: float = 32.0, force_collectives: bool = False):
        assert dist.is_initialized()
        self.group = group
        self.force_collectives = force_collectives          # tests: issue the collectives even in a group of one rank
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self._side = None

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
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                flat.record_stream(side)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
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
            dist.broadcast(tensor, src=src, group=self.group)
