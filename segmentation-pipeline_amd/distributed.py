"""Multi-GPU execution of the hot path: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for the plumbing tests).

The reference is single-process / single-device, so nothing here
mirrors reference code; it shards the two loops the reference runs serially:

* training (segmentation_trainer.py:162-180): patches are independent units, one
  micro-batch per rank, and the only exchange is the gradient.  `PatchParallel`
  packs gradients into a few flat fp32 buckets (one multi-tensor copy per bucket, as
  soon as the backward pass has produced all of its gradients, reverse parameter
  order) and launches one asynchronous all-reduce per bucket, so RCCL overlaps with
  the rest of the backward; afterwards every .grad is a view of its bucket.  18.08 M params = 72.3 MB -> 3 buckets of <= 25 MB; an xGMI ring moves
  2*(7/8)*25 MB per link per bucket, far below one backward pass.
* sliding-window inference (prediction.py:124-152): tiles are independent units;
  tile i goes to rank i % world and ONE all_gather returns the per-tile outputs, which
  rank order then aggregates in grid order, bit-identical to the 1-GPU result.
"""
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


# ---- sharding of independent inference units (sliding-window tiles, ensemble members) ----------
# Opt-in and exclusive.  Sharding assumes EVERY rank holds the same input and calls the predictor
# at the same time, which is not true for e.g. rank-local validation inside a DDP training job, so
# it must be requested (`with unit_sharding():`).  Only the OUTERMOST sharder inside such a region
# shards; nested ones (an ensemble of ensembles, research/msseg2/competition/ms-inference.py:115-125,
# or PatchPredict around an ensemble) run all of their units locally -- otherwise the ranks would
# issue different collective sequences (a hang on RCCL) or mix tiles of different members.
_shard_enabled = False
_shard_depth = 0


class unit_sharding:
    """`with unit_sharding():` -- PatchPredict / Ensemble* called inside spread their units over
    the ranks of the default process group (one gather per call).  No-op without torch.distributed."""

    def __init__(self, enabled: bool = True):
        self.enabled = enabled

    def __enter__(self):
        global _shard_enabled
        self.prev = _shard_enabled
        _shard_enabled = self.enabled
        return self

    def __exit__(self, *exc):
        global _shard_enabled
        _shard_enabled = self.prev


class shard_scope:
    """Used by the sharders themselves: `with shard_scope() as active:` -- `active` is True for the
    outermost sharder of an enabled region when torch.distributed has more than one rank."""

    def __enter__(self):
        global _shard_depth
        active = _shard_enabled and _shard_depth == 0 and is_distributed()
        _shard_depth += 1
        return active

    def __exit__(self, *exc):
        global _shard_depth
        _shard_depth -= 1


class PatchParallel(nn.Module):
    """Data-parallel wrapper with bucketed, backward-overlapped gradient all-reduce.

    Parameters that never receive a gradient (e.g. the unused `bias` of the reference's
    Blur / WS convolutions, components.py:86,119,152) are tolerated: their slice of the
    bucket stays zero and they keep `.grad is None`.

    sync_batch_norm: BatchNorm layers in training mode normalise with the statistics of the batch over ALL ranks
    (ops.batch_norm_sync: two per-channel all-reduces per layer and step), so a BatchNorm model
    (models/nested_residual_unet.py:19-23) trained on a sharded batch follows the same trajectory as the reference's
    single process on the whole batch.  Off: per-rank statistics, running buffers averaged (`sync_buffers`).
    """

    def __init__(self, module: nn.Module, bucket_bytes: int = 25 << 20, process_group=None,
                 broadcast_parameters: bool = True, force_collectives: bool = False, sync_batch_norm: bool = False,
                 tail_bucket_bytes: int = 4 << 20, bucket_dtype: torch.dtype = torch.float32):
        super().__init__()
        if bucket_dtype not in (torch.float32, torch.bfloat16):
            # float16 is refused: the loss scale of the fp16 mode is already removed where a parameter gradient is formed,
            # so the wire would carry TRUE gradients of a mean-type loss (1e-4 .. 1e-8): fp16 subnormals and zeros
            raise ValueError(f"bucket_dtype must be float32 or bfloat16 (same exponent range as the fp32 master "
                             f"gradients), not {bucket_dtype}")
        self.module = module
        self.tail_bucket_bytes = tail_bucket_bytes
        # bucket_dtype: the WIRE type of the gradient all-reduce.  float32 (default): the fp32 buckets themselves are
        # reduced.  bfloat16 (the 16-bit precision modes, SURVEY section 5: 36.2 MB instead of 72.3 MB per cfg2
        # step): each bucket is cast into a 16-bit wire buffer by the packing copy, reduced on the wire type and
        # cast back into the fp32 bucket the optimizer reads -- master gradients and weights stay fp32.
        self.bucket_dtype = bucket_dtype
        self.group = process_group
        self.sync_batch_norm = sync_batch_norm
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collectives: run the hooks / all-reduces even with one rank (exercises the RCCL path
        # on a single-GPU box; numerically a no-op)
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())
        self.params = [p for p in module.parameters() if p.requires_grad]
        self._build_buckets(bucket_bytes)
        self._pending: List = []
        self.hooks_enabled = True      # (trainer.SegmentedGraphTrainStep captures the backward with the hooks off)
        self._ready = [0] * len(self.buckets)
        self._used = [set() for _ in self.buckets]
        self._hooks = []
        # RCCL / NCCL average in the collective itself; gloo (CPU tests) needs an explicit division
        self._backend_has_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        if self.active:
            if broadcast_parameters:
                # into `detach()` (shares storage AND the version counter with the parameter), not `.data`: the
                # version-keyed caches (packed weights, derived filters, captured graphs) of a model that ran a
                # forward before being wrapped must see that ranks != 0 now hold other values
                with torch.no_grad():
                    for t in list(module.parameters()) + list(module.buffers()):
                        dist.broadcast(t.detach(), src=0, group=self.group)
            for idx, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(idx)))

    # -- buckets -------------------------------------------------------------
    def _build_buckets(self, bucket_bytes):
        """Reverse parameter order ~ the order in which backward produces gradients."""
        self.buckets = []       # flat fp32 tensors
        self.wire = []          # per bucket: the 16-bit wire buffer (bucket_dtype != float32), else None
        self.bucket_of = {}     # param index -> (bucket, offset)
        self.members = []       # per bucket: list of param indices
        cur, cur_bytes = [], 0
        order = list(reversed(range(len(self.params))))
        groups = []
        for idx in order:
            nbytes = self.params[idx].numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(idx)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        # The all-reduce of the LAST bucket (the first layers' gradients, produced at the very end of backward) cannot
        # overlap with anything: optimizer.step waits for it.  Its members are the smallest parameters of a U-Net, but
        # a 25 MB bucket also swallows the mid-level layers in front of them; peel the tail off into a bucket of its
        # own (<= tail_bucket_bytes: 3.5 MB for cfg2's d0..d2 convs) so only that much ring time is exposed.
        tail = getattr(self, "tail_bucket_bytes", 0)
        if groups and tail > 0:
            last, peeled, peeled_bytes = groups[-1], [], 0
            while len(last) > 1 and peeled_bytes + self.params[last[-1]].numel() * 4 <= tail:
                peeled_bytes += self.params[last[-1]].numel() * 4
                peeled.insert(0, last.pop())
            if peeled and sum(self.params[i].numel() * 4 for i in last) > tail:
                groups.append(peeled)
            else:
                last.extend(peeled)   # nothing to gain: the bucket was small anyway
        for b, idxs in enumerate(groups):
            total = sum(self.params[i].numel() for i in idxs)
            dev = self.params[idxs[0]].device
            flat = torch.zeros(total, dtype=torch.float32, device=dev)
            off = 0
            for i in idxs:
                self.bucket_of[i] = (b, off)
                off += self.params[i].numel()
            self.buckets.append(flat)
            wire_dtype = getattr(self, "bucket_dtype", torch.float32)
            self.wire.append(None if wire_dtype == torch.float32 else torch.zeros(total, dtype=wire_dtype, device=dev))
            self.members.append(idxs)

    def _grad_view(self, idx):
        b, off = self.bucket_of[idx]
        p = self.params[idx]
        return self.buckets[b][off:off + p.numel()].view_as(p)

    def _make_hook(self, idx):
        b, _ = self.bucket_of[idx]

        def hook(param):
            if not self.hooks_enabled:
                return
            # autograd has just stored this parameter's gradient (.grad was None, so it was not added
            # into anything: no kernel); count it, and ship the bucket when its last member arrives
            self._used[b].add(idx)
            self._ready[b] += 1
            if self._ready[b] == len(self.members[b]):
                self._launch(b)
        return hook

    def _launch(self, b):
        """Pack the gradients of bucket b with ONE multi-tensor copy and start its all-reduce."""
        idxs = [i for i in self.members[b] if self.params[i].grad is not None]
        views = [self._grad_view(i) for i in idxs]
        wire = self.wire[b]
        if wire is not None:
            # pack + cast in the same multi-tensor copy: gradient -> its slice of the 16-bit wire buffer
            wviews = []
            for i in idxs:
                _, off = self.bucket_of[i]
                wviews.append(wire[off:off + self.params[i].numel()].view_as(self.params[i]))
            if idxs:
                torch._foreach_copy_(wviews, [self.params[i].grad for i in idxs])
            if len(idxs) < len(self.members[b]):     # members without a gradient this step: their wire slice must not
                for i in self.members[b]:            # keep last step's reduced value
                    if self.params[i].grad is None:
                        _, off = self.bucket_of[i]
                        wire[off:off + self.params[i].numel()].zero_()
        else:
            src = [(v, self.params[i].grad) for v, i in zip(views, idxs) if self.params[i].grad.data_ptr() != v.data_ptr()]
            if src:
                torch._foreach_copy_([v for v, _ in src], [g for _, g in src])
        for v, i in zip(views, idxs):
            self.params[i].grad = v  # the optimizer reads the reduced (fp32) bucket
        # slices of parameters without a gradient are never written: they stay at their initial zero
        avg = self._backend_has_avg
        work = dist.all_reduce(self.buckets[b] if wire is None else wire, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM,
                               group=self.group, async_op=True)
        self._pending.append((b, work))

    # -- step protocol ---------------------------------------------------------
    def forward(self, *args, **kwargs):
        if self.sync_batch_norm and self.active:
            from . import ops
            with ops.batch_norm_sync(self.group if self.group is not None else dist.group.WORLD):
                return self.module(*args, **kwargs)
        return self.module(*args, **kwargs)

    def zero_grad(self, set_to_none: bool = True):
        """Drop the gradients (the buckets are overwritten by the next backward; nothing to memset).
        One backward pass per step: gradient accumulation over several passes is not supported."""
        for p in self.params:
            p.grad = None

    def finish_gradient_sync(self):
        """Call after loss.backward(): flushes buckets with unused parameters, waits for the
        collectives and turns the sums into means (gradient of the mean loss over ranks)."""
        if not self.active:
            return
        launched = {b for b, _ in self._pending}
        for b in range(len(self.buckets)):
            if b not in launched:
                self._launch(b)  # some members never produced a gradient this step
        for b, work in self._pending:
            work.wait()
            if self.wire[b] is not None:
                self.buckets[b].copy_(self.wire[b])      # 16-bit wire -> the fp32 master gradients (.grad views)
            if not self._backend_has_avg:
                self.buckets[b].div_(self.world)
        self._pending.clear()
        self._ready = [0] * len(self.buckets)
        self._used = [set() for _ in self.buckets]
        self.sync_buffers()

    def sync_buffers(self):
        """BatchNorm running statistics are updated from per-rank batches (the reference's
        per-process semantics, models/components.py:24): average the floating-point buffers over the
        ranks in ONE flat all-reduce so that eval-mode predictions and `state_dict()` agree on every
        rank (integer buffers such as num_batches_tracked advance identically and are left alone)."""
        if not self.active:
            return
        bufs = [b for b in self.module.buffers() if b.is_floating_point() and b.numel() > 0]
        if not bufs:
            return
        flat = torch.cat([b.detach().reshape(-1).float() for b in bufs])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat /= self.world
        off = 0
        with torch.no_grad():
            for b in bufs:
                b.detach().copy_(flat[off:off + b.numel()].view_as(b))   # (bumps the buffer's version: see __init__)
                off += b.numel()

    def state_dict(self, *args, **kwargs):
        return self.module.state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        return self.module.load_state_dict(*args, **kwargs)


def all_reduce_mean_scalars(values: torch.Tensor, group=None):
    """Loss-dict logging and the cooperative stop flag (segmentation_trainer.py:270-275):
    one small all-reduce so every rank leaves the loop on the same iteration."""
    if is_distributed():
        dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
        values /= dist.get_world_size(group)
    return values


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership: item i belongs to rank i % world."""
    return list(range(rank, n_items, world))


def gather_tiles(local_out: Optional[torch.Tensor], n_tiles: int, tile_shape, dtype, device, group=None,
                 sharded: bool = True, dst: Optional[int] = None):
    """ONE collective: every rank contributes its tiles (padded to the max per-rank count);
    returns [n_tiles, *tile_shape] in global tile order -- on every rank (all_gather), or with `dst` set only on
    that rank (gather; the others get None and receive nothing).  `sharded=False`: the caller computed every tile
    locally (no collective)."""
    world = dist.get_world_size(group) if (sharded and is_distributed()) else 1
    rank = dist.get_rank(group) if (sharded and is_distributed()) else 0
    per = (n_tiles + world - 1) // world
    # the tile buffer itself is returned / sent only when it already has the requested dtype: a model whose tile output
    # is not `dtype` (a 16-bit head) is cast first, so that every rank contributes the same element type and size --
    # ranks with a ragged share always send a `dtype` staging buffer
    if local_out is not None and local_out.dtype != dtype:
        local_out = local_out.to(dtype)
    if world == 1:      # nothing to exchange: the local tiles ARE the result (no staging copy)
        return local_out if local_out is not None else torch.zeros((0,) + tuple(tile_shape), dtype=dtype, device=device)
    n_local = len(shard_indices(n_tiles, rank, world))
    if n_local == per and local_out is not None and local_out.is_contiguous():
        send = local_out                        # every slot filled: send the tile buffer itself
    else:
        send = torch.zeros((per,) + tuple(tile_shape), dtype=dtype, device=device)
        if n_local:
            send[:n_local] = local_out
    if dst is None:
        recv = torch.empty((world * per,) + tuple(tile_shape), dtype=dtype, device=device)
        dist.all_gather_into_tensor(recv, send, group=group)
    else:
        parts = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, parts, dst=dst, group=group)
        if rank != dst:
            return None
        recv = torch.cat(parts, dim=0)
    recv = recv.view((world, per) + tuple(tile_shape))
    # global tile i sits at [i % world, i // world]
    idx = torch.arange(n_tiles, device=device)
    return recv[idx % world, idx // world]
