"""Test-time ensembles (mirror of segmentation_pipeline/models/ensemble.py:9-103).

The member forward passes run on the HIP kernels.  Nothing of the reference's
`x.permute(..).flip(..)` -> model -> `.flip(..).permute(inverse)` -> `torch.stack` -> `mean` /
`argmax, mode, one_hot` chain is materialised per member: the member input is ONE gather pass
(`m355_flip_permute`), and each prediction is consumed where it lies by `m355_ensemble_accumulate`, which reads it
THROUGH the inverse index transform into a running sum ('mean') or casts its argmax as a vote into a per-class
histogram ('majority'); `m355_ensemble_finalize` divides / picks the winner (ties: smallest class, as torch.mode
on the CPU) and writes the int64 one-hot mask the reference returns.  Inside
`distributed.unit_sharding()` the members are independent units: member e runs on rank e % world, ONE all_gather
returns the (already back-mapped) predictions in member order and every rank reduces them in that order, so the
result is bit-identical to the single-GPU ensemble.
"""
import ctypes as C
import itertools
from typing import Sequence

import torch
from torch import nn
import torch.nn.functional as F

_STRATEGIES = ('mean', 'majority')
_IDENTITY = ((0, 1, 2), 0)


def parse_strategy(strategy: str):
    if strategy not in _STRATEGIES:
        raise ValueError(f"Ensembling strategy must be one of {_STRATEGIES} not {strategy}")
    return strategy


def apply_strategy(predictions: Sequence[torch.Tensor], strategy: str):
    """(E, N, C, ...) stack -> 'mean' over E, or 'majority': argmax over C, mode over E,
    one-hot back to (N, C, ...) (reference :16-35).  Host-side restatement on a materialised list (CPU tensors,
    and the tests); the modules below stream the members through the device kernels instead."""
    stacked = torch.stack(list(predictions))
    if strategy == 'mean':
        return stacked.mean(dim=0)
    if strategy == 'majority':
        num_classes = stacked.shape[2]
        votes = stacked.argmax(dim=2)
        winner = torch.mode(votes, dim=0).values
        return F.one_hot(winner, num_classes=num_classes).moveaxis(-1, 1)
    raise RuntimeError(f"Invalid prediction strategy {strategy}")


def _i32x3(v):
    return (C.c_int32 * 3)(*[int(a) for a in v])


class _Reducer:
    """Running reduction of member predictions on the device (one accumulator, no stack)."""

    def __init__(self, strategy, canonical_shape, device):
        self.mode = 0 if strategy == 'mean' else 1
        self.shape = tuple(canonical_shape)          # (N, C, D, H, W) in the ORIGINAL orientation
        self.device = device
        self.acc = None
        self.members = 0

    def add(self, pred, perm=(0, 1, 2), flip_mask=0):
        """pred: the member's prediction in the member's own orientation (fp32, contiguous)."""
        from .. import _lib, ops
        L = _lib.lib()
        N, Cc = self.shape[:2]
        if self.acc is None:
            self.acc = torch.empty(self.shape, dtype=torch.float32 if self.mode == 0 else torch.int32, device=self.device)
        pred = pred.contiguous()
        ops._require(pred)
        _lib.check(L.m355_ensemble_accumulate(ops._p(pred), ops._p(self.acc) if self.mode == 0 else None,
                                              ops._p(self.acc) if self.mode == 1 else None, N, Cc, _i32x3(self.shape[2:]),
                                              _i32x3(perm), int(flip_mask), self.mode, 1 if self.members == 0 else 0,
                                              ops._stream()), "ensemble_accumulate")
        self.members += 1

    def result(self):
        from .. import _lib, ops
        L = _lib.lib()
        N, Cc = self.shape[:2]
        S = self.shape[2] * self.shape[3] * self.shape[4]
        if self.mode == 0:
            out = torch.empty(self.shape, dtype=torch.float32, device=self.device)
            _lib.check(L.m355_ensemble_finalize(ops._p(self.acc), None, ops._p(out), None, N, Cc, S, self.members, 0,
                                                ops._stream()), "ensemble_finalize")
            return out, None
        out = torch.empty(self.shape, dtype=torch.int64, device=self.device)
        _lib.check(L.m355_ensemble_finalize(None, ops._p(self.acc), None, ops._p(out), N, Cc, S, self.members, 1,
                                            ops._stream()), "ensemble_finalize")
        return out, self.acc


def _member_input(x, perm, flip_mask):
    """x.permute(0, 1, *perm).flip(f).contiguous() as ONE gather pass."""
    if tuple(perm) == (0, 1, 2) and flip_mask == 0:
        return x
    from .. import _lib, ops
    x = x.contiguous()
    ops._require(x)
    N, Cc = x.shape[:2]
    sp = x.shape[2:]
    y = torch.empty((N, Cc) + tuple(sp[p] for p in perm), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().m355_flip_permute(ops._p(x), ops._p(y), N, Cc, _i32x3(sp), _i32x3(perm), int(flip_mask),
                                            ops._stream()), "flip_permute")
    return y


def _torch_member(x, perm, flip_mask):
    """CPU tensors (host-side tests of the sharding logic with a stock torch member): torch index ops."""
    dims = [2 + j for j in range(3) if (flip_mask >> j) & 1]
    y = x.permute(0, 1, *[2 + p for p in perm])
    return y.flip(dims) if dims else y


def _torch_back(y, perm, flip_mask):
    dims = [2 + j for j in range(3) if (flip_mask >> j) & 1]
    y = y.flip(dims) if dims else y
    inverse = [0, 0, 0]
    for j, p in enumerate(perm):
        inverse[p] = j
    return y.permute(0, 1, *[2 + j for j in inverse])


def _run_ensemble(module, x, members, strategy):
    """members: list of (model, perm, flip_mask).  -> the ensembled prediction (and module.last_votes for 'majority').
    Inside `distributed.unit_sharding()` the OUTERMOST ensemble (or sliding-window predictor) spreads its members
    over the ranks; nested ones run all of theirs locally (distributed.shard_scope)."""
    from .. import distributed as D
    on_device = x.is_cuda
    with D.shard_scope() as sharded:
        if not sharded and on_device:
            red = None
            for model, perm, fm in members:
                pred = model(_member_input(x, perm, fm))
                if red is None:
                    canon = (pred.shape[0], pred.shape[1]) + tuple(x.shape[2:])
                    red = _Reducer(strategy, canon, x.device)
                red.add(pred.float() if pred.dtype != torch.float32 else pred, perm, fm)
            out, votes = red.result()
            module.last_votes = votes
            return out
        # host path (CPU tensors) and the sharded path: predictions mapped back to the canonical orientation
        def run(i):
            model, perm, fm = members[i]
            if on_device:
                pred = model(_member_input(x, perm, fm))
                back = torch.empty((pred.shape[0], pred.shape[1]) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
                red = _Reducer('mean', back.shape, x.device)
                red.acc = back                     # one accumulate with first=1 IS the inverse transform
                # (a 'majority' member returns the reference's int64 one-hot mask: ms-inference.py:115-125 nests them)
                red.add(pred if pred.dtype == torch.float32 else pred.float(), perm, fm)
                return back
            return _torch_back(model(_torch_member(x, perm, fm)), perm, fm).contiguous()
        if not sharded:
            preds = [run(i) for i in range(len(members))]
        else:
            import torch.distributed as dist
            world, rank = dist.get_world_size(), dist.get_rank()
            mine = D.shard_indices(len(members), rank, world)
            local = [run(i) for i in mine]
            # every rank learns the member shape from rank 0 (which always owns member 0); ranks without a member
            # (fewer members than ranks) still take part, with tensors on the INPUT's device
            meta = torch.zeros(8, dtype=torch.int64, device=x.device)
            if rank == 0:
                meta[0] = local[0].dim()
                meta[1:1 + local[0].dim()] = torch.tensor(local[0].shape)
            dist.broadcast(meta, src=0)
            shape = tuple(int(v) for v in meta[1:1 + int(meta[0])])
            stacked = torch.stack(local) if local else None
            preds = list(D.gather_tiles(stacked, len(members), shape, torch.float32, x.device))
    if on_device:
        red = _Reducer(strategy, preds[0].shape, x.device)
        for p in preds:
            red.add(p)
        out, votes = red.result()
        module.last_votes = votes
        return out
    if strategy == 'majority':
        votes = torch.stack(preds).argmax(dim=2)
        module.last_votes = F.one_hot(votes, preds[0].shape[1]).sum(dim=0).moveaxis(-1, 1).to(torch.int32)
    return apply_strategy(preds, strategy)


def _flip_sets(dims):
    out = []
    for order in range(len(dims) + 1):
        out += list(itertools.combinations(dims, order))
    return out


def _mask(flip_dims):
    """flip dims given as tensor dims (2, 3, 4) -> bit j for spatial axis j"""
    m = 0
    for d in flip_dims:
        if d not in (2, 3, 4):
            raise NotImplementedError(f"ensemble flips act on the spatial dims (2, 3, 4), not {d}")
        m |= 1 << (d - 2)
    return m


class EnsembleModels(nn.Module):
    def __init__(self, models: Sequence[nn.Module], strategy: str = 'mean'):
        super().__init__()
        self.models = nn.ModuleList(models)
        self.strategy = parse_strategy(strategy)
        self.last_votes = None

    def forward(self, x):
        return _run_ensemble(self, x, [(m,) + _IDENTITY for m in self.models], self.strategy)


class EnsembleFlips(nn.Module):
    def __init__(self, model: nn.Module, strategy: str = 'mean', spatial_dims: Sequence[int] = (2, 3, 4)):
        super().__init__()
        self.model = model
        self.strategy = parse_strategy(strategy)
        self.spatial_dims = spatial_dims
        self.flips = _flip_sets(tuple(spatial_dims))
        self.last_votes = None

    def forward(self, x):
        return _run_ensemble(self, x, [(self.model, (0, 1, 2), _mask(f)) for f in self.flips], self.strategy)


class EnsembleOrientations(nn.Module):
    def __init__(self, model: nn.Module, strategy: str = 'mean'):
        super().__init__()
        self.model = model
        self.strategy = parse_strategy(strategy)
        dims = (2, 3, 4)
        self.permutations = list(itertools.permutations(dims))
        self.flips = _flip_sets(dims)
        self.last_votes = None

    def forward(self, x):
        members = [(self.model, tuple(d - 2 for d in perm), _mask(f)) for perm in self.permutations for f in self.flips]
        return _run_ensemble(self, x, members, self.strategy)
