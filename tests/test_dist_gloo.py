"""CPU, world_size 2, gloo: the data-parallel layer of the G+D step (stylegan/pytorch_amd/dist.py).

The kernels need the GPU, so the per-rank arithmetic here is the CPU oracle; what is under test is the N>1 logic
itself: the stddev-preserving shard, the SUM all-reduce with bucketing, the mean-vs-sum loss scaling
(softplus terms are batch means, R1 is a batch sum -- reference models/Losses.py:210,218), and the W-average
broadcast.  Target: N-rank gradients == single-process gradients at the GLOBAL batch."""
import datetime
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as TF

import golden_util as gu
from oracle import stylegan_oracle as O
from stylegan.pytorch_amd.dist import (BucketScheduler, DataParallelGroup, GradBuckets, bucketize, install_grad_hooks, note_grad_write,
                                       set_active_scheduler, stddev_preserving_shard)

WORLD = 2
RES, DEPTH_TOTAL, DEPTH, ALPHA, B = 16, 3, 2, 0.5, 16      # tiny D: 16x16, 8 channels


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _reap(procs):
    """A rank that died leaves its peer waiting in a collective: never leave a child behind."""
    for p in procs:
        if p.is_alive():
            p.terminate(); p.join(timeout=10)


def tiny_d_params():
    dp = O.make_discriminator_params(RES, fmap_base=32, fmap_max=8, dtype=torch.float64)
    for k in list(dp):
        dp[k] = gu.fill_value(k, dp[k].shape, torch.float64).requires_grad_(True)
    return dp


def local_d_loss(dp, real, fake, mean_scale):
    r = O.discriminator(dp, real, DEPTH, ALPHA, DEPTH_TOTAL)
    f = O.discriminator(dp, fake, DEPTH, ALPHA, DEPTH_TOTAL)
    loss = (TF.softplus(f).mean() + TF.softplus(-r).mean()) * mean_scale
    return loss + O.r1_penalty(dp, real, DEPTH, ALPHA, DEPTH_TOTAL) * 5.0


def test_shard_keeps_stddev_groups_whole():
    for bsz, world in [(8, 2), (32, 8), (16, 4), (16, 1)]:
        x = gu.seeded((bsz, 6, 4, 4), 3, torch.float64)
        full = O.minibatch_stddev(x)[:, -1]
        seen = []
        for rank in range(world):
            idx = stddev_preserving_shard(bsz, world, rank)
            seen += idx
            local = O.minibatch_stddev(x[idx])[:, -1]
            assert torch.allclose(local, full[idx], atol=1e-14), (bsz, world, rank)
        assert sorted(seen) == list(range(bsz))
    with pytest.raises(AssertionError):
        stddev_preserving_shard(12, 2, 0)                 # 12/4 = 3 groups-slots cannot be split over 2 ranks


def test_bucketize():
    assert bucketize([5, 5, 5, 20, 1], 10) == [[0, 1], [2], [3], [4]]
    assert bucketize([], 10) == []
    assert bucketize([100], 10) == [[0]]


def _worker(rank, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD, timeout=datetime.timedelta(seconds=120))
    torch.set_num_threads(2)
    try:
        group = DataParallelGroup(bucket_mb=0.001)          # tiny buckets: exercise the multi-bucket path
        dp = tiny_d_params()
        real = gu.seeded((B, 3, RES, RES), 1, torch.float64); fake = gu.seeded((B, 3, RES, RES), 2, torch.float64)
        idx = stddev_preserving_shard(B, WORLD, rank)
        loss = local_d_loss(dp, real[idx], fake[idx], mean_scale=1.0 / WORLD)
        names = sorted(dp)
        grads = torch.autograd.grad(loss, [dp[k] for k in names], allow_unused=True)   # unused from_rgb: grad None
        params = []
        for k, g in zip(names, grads):
            p = torch.nn.Parameter(dp[k].detach().clone()); p.grad = None if g is None else g.clone(); params.append(p)
        extra = torch.nn.Parameter(torch.zeros(3))           # inactive resolution: grad None on every rank, skipped
        group.all_reduce_grads(params + [extra])
        assert extra.grad is None
        # flat gradient buckets (what StyleGAN(data_parallel=...) uses from the second iteration at a depth on): the gradients
        # live in the buckets, the all-reduce runs on the buckets in place
        active = [p for p in params if p.grad is not None]
        gb = GradBuckets(active, group.bucket_elems)
        assert len(gb.buckets) > 1 and gb.matches(active) and not gb.matches(active[:-1])
        want = [p.grad.clone() for p in active]                          # (already the all-reduced sums)
        gb.attach()
        assert gb.attached() and all(float(p.grad.abs().sum()) == 0.0 for p in active)
        for p, w in zip(active, want):
            p.grad.add_(w * (0.25 if rank == 0 else 0.75))               # a "backward" accumulating into the views
        group.all_reduce_buckets(gb)
        for p, w in zip(active, want):
            assert torch.allclose(p.grad, w, rtol=1e-12, atol=0), "bucket all-reduce"
        loss_part = torch.tensor(1.5 + rank, dtype=torch.float64)
        assert float(group.all_reduce_scalar(loss_part)) == 4.0 and float(loss_part) == 1.5 + rank    # a new tensor; the input is untouched
        # host decisions every rank must take identically (round 6: launch mode of a depth from the slowest rank's timings; "did every
        # rank's hipGraph capture succeed" -- a rank replaying [graph | all-reduce | update] next to an eager rank would hang the group)
        assert group.host_max([1.0 + rank, 5.0 - rank]) == [2.0, 5.0]
        assert group.all_ok(True) is True and group.all_ok(rank == 0) is False and group.all_ok(False) is False
        avg = torch.full((4,), float(rank + 1))
        group.broadcast(avg, src=0)
        # numpy payloads are pickled by value (torch tensors would travel as shared-memory handles of a dying process)
        if rank == 0:
            out_q.put(({k: (None if p.grad is None else p.grad.numpy()) for k, p in zip(names, params)}, avg.numpy()))
        else:
            out_q.put(avg.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gradients_equal_global_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q), daemon=True) for r in range(WORLD)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=120) for _ in range(WORLD)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        _reap(procs)
    grads = next(g for g in got if isinstance(g, tuple))[0]
    for g in got:
        avg = g[1] if isinstance(g, tuple) else g
        assert (torch.as_tensor(avg) == 1.0).all()           # rank 0's buffer everywhere
    # single process, global batch, un-scaled loss
    dp = tiny_d_params()
    real = gu.seeded((B, 3, RES, RES), 1, torch.float64); fake = gu.seeded((B, 3, RES, RES), 2, torch.float64)
    loss = local_d_loss(dp, real, fake, mean_scale=1.0)
    names = sorted(dp)
    ref = dict(zip(names, torch.autograd.grad(loss, [dp[k] for k in names], allow_unused=True)))
    assert any(v is None for v in ref.values())
    for k in names:
        if ref[k] is None:
            assert grads[k] is None
            continue
        err = (torch.as_tensor(grads[k]) - ref[k]).abs().max().item()
        assert err <= 1e-10 * (ref[k].abs().max().item() + 1e-30) + 1e-14, (k, err)


# ---- every loss head under data parallelism (the product's Losses classes; the discriminator is a small fp64 stand-in) ----

class _TinyD(torch.nn.Module):
    """[B,3,8,8] (+ labels) -> [B,1]; same call signature as the product's Discriminator."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(11)
        self.w1 = torch.nn.Parameter(torch.randn(3 * 8 * 8, 12, generator=g, dtype=torch.float64) * 0.2)
        self.w2 = torch.nn.Parameter(torch.randn(12, 1, generator=g, dtype=torch.float64))
        self.emb = torch.nn.Parameter(torch.randn(4, 12, generator=g, dtype=torch.float64) * 0.5)

    def forward(self, x, height, alpha, labels_in=None):
        h = TF.leaky_relu(x.flatten(1) @ self.w1, 0.2)
        if labels_in is not None:
            h = h + self.emb[labels_in]
        return h @ self.w2 * (1.0 + alpha) + 0.3


LOSS_HEADS = ["logistic", "hinge", "standard-gan", "relativistic-hinge", "conditional-loss"]


def _head(name, dis, mean_scale, group):
    from stylegan.pytorch_amd import Losses
    if name == "logistic":
        # the product evaluates the softplus terms in one HIP launch and has no host path: on the CPU this test stands in for that
        # kernel with the formula it implements (reference models/Losses.py:216-218,226) -- what is under test here is the class's
        # data-parallel scaling around it
        def _cpu_logistic_heads(f_preds, r_preds, ms, gen):
            if gen:
                return torch.mean(TF.softplus(-f_preds)) * ms
            return (torch.mean(TF.softplus(f_preds)) + torch.mean(TF.softplus(-r_preds))) * ms
        Losses.logistic_heads = _cpu_logistic_heads
        return Losses.LogisticGAN(dis, mean_scale=mean_scale)
    if name == "hinge":
        return Losses.HingeGAN(dis, mean_scale=mean_scale)
    if name == "standard-gan":
        return Losses.StandardGAN(dis, mean_scale=mean_scale)
    if name == "relativistic-hinge":
        return Losses.RelativisticAverageHingeGAN(dis, mean_scale=mean_scale,
                                                  batch_mean=group.global_mean if group is not None else None)
    return Losses.ConditionalGANLoss(dis, mean_scale=mean_scale)


def _head_losses(name, dis, real, fake, labels, mean_scale, group):
    head = _head(name, dis, mean_scale, group)
    if name == "conditional-loss":
        return head.dis_loss(real, fake, labels, 2, 0.5), head.gen_loss(real, fake, labels, 2, 0.5)
    if name == "logistic":                                   # (the R1 term needs the product's kernels: GPU test)
        return head.dis_loss(real, fake, 2, 0.5, r1_gamma=0.0), head.gen_loss(real, fake, 2, 0.5)
    return head.dis_loss(real, fake, 2, 0.5), head.gen_loss(real, fake, 2, 0.5)


def _head_inputs():
    real = gu.seeded((B, 3, 8, 8), 21, torch.float64); fake = gu.seeded((B, 3, 8, 8), 22, torch.float64)
    labels = torch.arange(B) % 4
    return real, fake, labels


def _head_worker(rank, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD, timeout=datetime.timedelta(seconds=120))
    torch.set_num_threads(2)
    try:
        group = DataParallelGroup()
        real, fake, labels = _head_inputs()
        idx = stddev_preserving_shard(B, WORLD, rank)
        out = {}
        for name in LOSS_HEADS:
            for which in (0, 1):
                dis = _TinyD()
                loss = _head_losses(name, dis, real[idx], fake[idx], labels[idx], 1.0 / WORLD, group)[which]
                loss.backward()
                group.all_reduce_grads(dis.parameters())
                total = group.all_reduce_scalar(loss)
                out[(name, which)] = (float(total), {k: None if p.grad is None else p.grad.numpy()
                                                     for k, p in dis.named_parameters()})
        if rank == 0:
            out_q.put(out)
    finally:
        dist.destroy_process_group()


def test_every_loss_head_two_ranks_equal_global_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_head_worker, args=(r, port, q), daemon=True) for r in range(WORLD)]
    for p in procs:
        p.start()
    try:
        got = q.get(timeout=120)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        _reap(procs)
    real, fake, labels = _head_inputs()
    for name in LOSS_HEADS:
        for which in (0, 1):
            dis = _TinyD()
            loss = _head_losses(name, dis, real, fake, labels, 1.0, None)[which]
            loss.backward()
            total, grads = got[(name, which)]
            want = loss.item()
            assert abs(total - want) <= 1e-12 * max(1.0, abs(want)), (name, which, total, want)
            for k, p in dis.named_parameters():
                if p.grad is None:
                    assert grads[k] is None, (name, which, k)
                    continue
                err = (torch.as_tensor(grads[k]) - p.grad).abs().max().item()
                assert err <= 1e-12 * (p.grad.abs().max().item() + 1e-30) + 1e-15, (name, which, k, err)


def test_relativistic_mean_must_be_global():
    """The same comparison with the LOCAL mean in the relativistic head is off: the test above is sensitive to the wiring."""
    real, fake, labels = _head_inputs()
    dis = _TinyD()
    want = _head_losses("relativistic-hinge", dis, real, fake, labels, 1.0, None)[0].item()
    part = 0.0
    for rank in range(WORLD):
        idx = stddev_preserving_shard(B, WORLD, rank)
        part += _head_losses("relativistic-hinge", dis, real[idx], fake[idx], labels[idx], 1.0 / WORLD, None)[0].item()
    assert abs(part - want) > 1e-6


# ---- bucket-level overlap: a bucket is all-reduced as soon as its last gradient of the backward is final ----
def _overlap_worker(rank, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD, timeout=datetime.timedelta(seconds=120))
    torch.set_num_threads(2)
    try:
        group = DataParallelGroup(bucket_mb=0.001)          # tiny buckets: several per network
        # (1) autograd path: the oracle's D step on this rank's shard, gradients through AccumulateGrad + post-accumulate hooks
        dp = tiny_d_params()
        real = gu.seeded((B, 3, RES, RES), 1, torch.float64); fake = gu.seeded((B, 3, RES, RES), 2, torch.float64)
        idx = stddev_preserving_shard(B, WORLD, rank)
        leaves = list(dp.values())
        install_grad_hooks(leaves)
        sched = BucketScheduler(group)                      # iteration 1: record
        set_active_scheduler(sched)
        local_d_loss(dp, real[idx], fake[idx], mean_scale=1.0 / WORLD).backward()
        set_active_scheduler(None)
        active = [p for p in leaves if p.grad is not None]
        assert len(active) < len(leaves)                    # the inactive resolutions never got a note
        gb = sched.layout(group.bucket_elems, only=active, canonical=leaves)
        assert sched.order_source == "rank0"
        assert gb is not None and gb.matches(active) and len(gb.buckets) > 2
        first_iter = {k: v.grad.clone() for k, v in dp.items() if v.grad is not None}
        gb.attach()                                         # iteration 2: the gradients live in the buckets, zeroed
        sched.begin()
        set_active_scheduler(sched)
        fired_during = []
        orig_fire = sched._fire
        sched._fire = lambda b: (fired_during.append(b), orig_fire(b))[1]
        local_d_loss(dp, real[idx], fake[idx], mean_scale=1.0 / WORLD).backward()
        set_active_scheduler(None)
        n_early = len(fired_during)
        sched.finish()
        assert n_early == len(gb.buckets) == sched.fired_early      # every bucket left during the backward, none at the join
        assert fired_during == sorted(fired_during)          # ... in layout (= gradient-ready) order
        for k, g in first_iter.items():                      # same local gradients both times; now summed over the ranks
            assert dp[k].grad.shape == g.shape
        # (2) several contributions per parameter, noted by hand (what ConvFn.backward does for in-kernel accumulation)
        ps = [torch.nn.Parameter(torch.zeros(n, dtype=torch.float64)) for n in (300, 7, 1200, 64, 500)]
        contrib = {0: 3, 1: 1, 2: 2, 3: 3, 4: 1}
        # write order: on rank 0 parameter 1 is final first, then 2, 0, 4, 3.  Rank 1 is deliberately PERTURBED (same counts,
        # another order: final 4, 3, 1, 0, 2) -- what autograd's thread-local sequence numbers can do to the order in which a
        # parameter's contributions are summed.  The layout must still be rank 0's on both ranks and the collectives must match.
        seq = [0, 2, 3, 0, 1, 3, 2, 0, 4, 3] if rank == 0 else [3, 3, 0, 4, 2, 0, 3, 1, 0, 2]
        def val(i, j): return torch.full_like(ps[i], float((rank + 1) * (10 * i + j + 1)))
        def backward(sch):
            seen = {i: 0 for i in contrib}
            for i in seq:
                g = val(i, seen[i]); seen[i] += 1
                ps[i].grad = g.clone() if ps[i].grad is None else ps[i].grad.add_(g)
                sch.note(ps[i])
        s2 = BucketScheduler(group)
        backward(s2)
        local_order = [next(i for i in contrib if id(ps[i]) == k) for k in s2.order]
        assert local_order == ([1, 2, 0, 4, 3] if rank == 0 else [4, 3, 1, 0, 2])
        gb2 = s2.layout(group.bucket_elems, canonical=ps)
        assert [id(p) for p in gb2.params] == [id(ps[i]) for i in (1, 2, 0, 4, 3)]      # RANK 0's gradient-ready order, on every rank
        assert s2.order_source == "rank0" and len(gb2.buckets) == 5                       # five buckets of five different sizes
        fired2 = []
        fire2 = s2._fire
        s2._fire = lambda b: (fired2.append(b), fire2(b))[1]
        gb2.attach(); s2.begin(); backward(s2)
        assert sum(s2.fired) == len(gb2.buckets)
        assert fired2 == list(range(len(gb2.buckets)))       # fired in BUCKET order on both ranks, whatever the local completion order
        s2.finish()
        for i in contrib:
            want = sum(float((r + 1) * (10 * i + j + 1)) for r in range(WORLD) for j in range(contrib[i]))
            assert torch.allclose(ps[i].grad, torch.full_like(ps[i], want)), (i, float(ps[i].grad[0]), want)
        # a write beyond the recorded count means a bucket left too early: refused loudly
        gb2.attach(); s2.begin(); backward(s2); s2.note(ps[1])
        try:
            s2.finish(); raised = False
        except RuntimeError:
            raised = True
        assert raised
        # ranks that recorded different contribution COUNTS (or sets) must not build a layout: refused on every rank together
        s3 = BucketScheduler(group)
        for i in ([0, 1, 0, 2] if rank == 0 else [0, 1, 2, 2]):
            s3.note(ps[i])
        try:
            s3.layout(group.bucket_elems, canonical=ps); raised = False
        except RuntimeError:
            raised = True
        assert raised and s3.recording
        if rank == 0:
            out_q.put({k: v.grad.numpy() for k, v in dp.items() if v.grad is not None})
        else:
            out_q.put(None)
    finally:
        dist.destroy_process_group()


def test_buckets_are_reduced_as_soon_as_their_gradients_are_final():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, port, q), daemon=True) for r in range(WORLD)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=120) for _ in range(WORLD)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        _reap(procs)
    grads = next(g for g in got if g is not None)
    dp = tiny_d_params()
    real = gu.seeded((B, 3, RES, RES), 1, torch.float64); fake = gu.seeded((B, 3, RES, RES), 2, torch.float64)
    loss = local_d_loss(dp, real, fake, mean_scale=1.0)
    names = sorted(dp)
    ref = dict(zip(names, torch.autograd.grad(loss, [dp[k] for k in names], allow_unused=True)))
    assert sorted(grads) == sorted(k for k in names if ref[k] is not None)
    for k, g in grads.items():
        err = (torch.as_tensor(g) - ref[k]).abs().max().item()
        assert err <= 1e-10 * (ref[k].abs().max().item() + 1e-30) + 1e-14, (k, err)
