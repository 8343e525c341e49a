"""Test-time ensembles (mirror of segmentation_pipeline/models/ensemble.py:9-103).

The member forward passes run on the HIP kernels; the reductions over the ensemble
axis are index / selection arithmetic on the stacked predictions.  Inside
`distributed.unit_sharding()` the members are independent units: member e runs on rank
e % world and ONE all_gather returns the predictions in member order, so the result is
bit-identical to the single-GPU ensemble.
"""
import itertools
from typing import Sequence

import torch
from torch import nn
import torch.nn.functional as F

_STRATEGIES = ('mean', 'majority')


def parse_strategy(strategy: str):
    if strategy not in _STRATEGIES:
        raise ValueError(f"Ensembling strategy must be one of {_STRATEGIES} not {strategy}")
    return strategy


def apply_strategy(predictions: Sequence[torch.Tensor], strategy: str):
    """(E, N, C, ...) stack -> 'mean' over E, or 'majority': argmax over C, mode over E,
    one-hot back to (N, C, ...) (reference :16-35)."""
    stacked = torch.stack(list(predictions))
    if strategy == 'mean':
        return stacked.mean(dim=0)
    if strategy == 'majority':
        num_classes = stacked.shape[2]
        votes = stacked.argmax(dim=2)
        winner = torch.mode(votes, dim=0).values
        return F.one_hot(winner, num_classes=num_classes).moveaxis(-1, 1)
    raise RuntimeError(f"Invalid prediction strategy {strategy}")


def _run_members(thunks, device):
    """Evaluate the member forward passes; inside `distributed.unit_sharding()` the OUTERMOST
    ensemble (or sliding-window predictor) spreads its members over the ranks, nested ones run all
    of theirs locally (see distributed.shard_scope)."""
    from .. import distributed as D
    with D.shard_scope() as sharded:
        if not sharded:
            return [t() for t in thunks]
        import torch.distributed as dist
        world, rank = dist.get_world_size(), dist.get_rank()
        mine = D.shard_indices(len(thunks), rank, world)
        local = [thunks[i]().contiguous() for i in mine]
        # every rank learns the member shape from rank 0 (which always owns member 0); ranks without a
        # member (fewer members than ranks) still take part, with tensors on the INPUT's device
        meta = torch.zeros(8, dtype=torch.int64, device=device)
        if rank == 0:
            meta[0] = local[0].dim()
            meta[1:1 + local[0].dim()] = torch.tensor(local[0].shape)
        dist.broadcast(meta, src=0)
        shape = tuple(int(v) for v in meta[1:1 + int(meta[0])])
        stacked = torch.stack(local) if local else None
        out = D.gather_tiles(stacked, len(thunks), shape, torch.float32, device)
        return list(out)


def _flip_sets(dims):
    out = []
    for order in range(len(dims) + 1):
        out += list(itertools.combinations(dims, order))
    return out


class EnsembleModels(nn.Module):
    def __init__(self, models: Sequence[nn.Module], strategy: str = 'mean'):
        super().__init__()
        self.models = nn.ModuleList(models)
        self.strategy = parse_strategy(strategy)

    def forward(self, x):
        return apply_strategy(_run_members([(lambda m=m: m(x)) for m in self.models], x.device), self.strategy)


class EnsembleFlips(nn.Module):
    def __init__(self, model: nn.Module, strategy: str = 'mean', spatial_dims: Sequence[int] = (2, 3, 4)):
        super().__init__()
        self.model = model
        self.strategy = parse_strategy(strategy)
        self.spatial_dims = spatial_dims
        self.flips = _flip_sets(tuple(spatial_dims))

    def forward(self, x):
        thunks = [(lambda f=f: self.model(x.flip(f).contiguous()).flip(f)) for f in self.flips]
        return apply_strategy(_run_members(thunks, x.device), self.strategy)


class EnsembleOrientations(nn.Module):
    def __init__(self, model: nn.Module, strategy: str = 'mean'):
        super().__init__()
        self.model = model
        self.strategy = parse_strategy(strategy)
        dims = (2, 3, 4)
        self.permutations = list(itertools.permutations(dims))
        self.flips = _flip_sets(dims)

    def forward(self, x):
        thunks = []
        for perm in self.permutations:
            inverse = tuple((torch.argsort(torch.tensor(perm)) + 2).tolist())
            for f in self.flips:
                thunks.append(lambda perm=perm, inverse=inverse, f=f:
                              self.model(x.permute(0, 1, *perm).flip(f).contiguous()).flip(f).permute(0, 1, *inverse))
        return apply_strategy(_run_members(thunks, x.device), self.strategy)
