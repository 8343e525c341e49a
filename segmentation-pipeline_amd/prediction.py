"""Tensor-level predictors: the arithmetic of segmentation_pipeline/prediction.py
(reference) without the torchio Subject plumbing.

* split_and_flip / reverse_split_and_flip (:16-27): index-only views.
* StandardPredict.predict (:73-102): optional sagittal split, one model call; the
  prediction stays on the device (the reference's per-subject `.detach().cpu()`, :97,
  is an evaluator concern and forces a sync every iteration).
* PatchPredict.predict (:124-152): torchio GridSampler / GridAggregator('average')
  become a deterministic tile list + the patch_gather / patch_accumulate kernels;
  inside `distributed.unit_sharding()` the tiles are sharded over the ranks and
  returned by a single all_gather (distributed.gather_tiles).
"""
import itertools
from typing import Optional, Sequence, Tuple

import torch

from . import distributed as D
from . import ops


def split_and_flip(x: torch.Tensor) -> torch.Tensor:
    first, second = x.split(x.shape[2] // 2, dim=2)
    return torch.cat([first, second.flip(2)], dim=0)


def reverse_split_and_flip(x: torch.Tensor) -> torch.Tensor:
    first, second = x.split(x.shape[0] // 2, dim=0)
    return torch.cat([first, second.flip(2)], dim=2)


def grid_locations(volume_shape, patch_size, patch_overlap):
    """Corner indices of torchio 0.18.45's GridSampler with padding_mode=None (the
    configuration of research/msseg2/msseg2.py:139-146): per axis
    range(0, size - patch + 1, patch - overlap), plus size - patch if the border is not
    reached; patches enumerated in itertools.product order."""
    return [tuple(c) for c in itertools.product(*grid_axes(volume_shape, patch_size, patch_overlap))]


def grid_axes(volume_shape, patch_size, patch_overlap):
    """the per-axis start lists whose product is `grid_locations`"""
    axes = []
    for size, p, o in zip(volume_shape, patch_size, patch_overlap):
        if p > size:
            raise ValueError(f"patch size {p} exceeds volume size {size}")
        if o >= p:
            raise ValueError(f"patch overlap {o} must be smaller than patch size {p}")
        starts = list(range(0, size - p + 1, p - o))
        if starts[-1] != size - p:
            starts.append(size - p)
        axes.append(starts)
    return axes


def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


class StandardPredict:
    """Whole-image prediction (reference :57-102), tensors in, tensors out."""

    def __init__(self, image_names: Sequence[str] = ("X",), sagittal_split: bool = False, refine_image: str = None):
        image_names = list(image_names)
        if refine_image is not None and refine_image not in image_names:
            image_names.append(refine_image)
        self.image_names = image_names
        self.sagittal_split = sagittal_split
        self.refine_image = refine_image

    def predict(self, model, device, batch, label_attributes=None):
        """batch: dict of stacked tensors (collate_subjects, utils/utils.py:75-85)."""
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        x = batch["X"]
        if self.sagittal_split:
            y_pred = reverse_split_and_flip(model(split_and_flip(x).contiguous()))
        else:
            y_pred = model(x)
        batch["y_pred"] = y_pred
        return batch


class GraphedForward:
    """The no-grad forward of `model` replayed from a hipGraph, one capture per (input shape, precision mode, parameter
    versions).  A sliding window pushes hundreds of same-shaped tile batches through the model (reference :136-141);
    with small patches the forward is launch-bound -- ~80 launches of a few microseconds of GPU work each against
    1.3-2 ms of host enqueue time -- and a replay is one launch.  Capture is safe here because every launch of the
    library goes to the current stream, the work queues are reset on the device, workspaces come from torch's
    allocator (the graph's private pool) and the packed-weight caches are filled by the warm-up passes; the result
    is bit-identical to the eager forward (`tools/graph_probe.py`, tests).  The output tensor is owned by the graph:
    it is valid until the next call with the same key (callers that keep it, as PatchPredict does, get a copy).
    Large patches are GPU-bound and gain nothing (cfg4: 2.14 ms eager vs 2.21 ms replayed in bf16)."""

    def __init__(self, model, copy_output: bool = True, max_graphs: int = 4):
        self.model, self.copy_output, self.max_graphs = model, copy_output, max_graphs
        self._graphs = {}

    def _key(self, x):
        # storage AND version of every parameter / buffer: a replaced Parameter, `module.to()` or an optimizer step
        # all invalidate the capture (a sum of versions alone would replay a graph that reads freed storage).
        # In-place writes through `.data` bump no version and are not supported on a captured model.
        version = hash(tuple((t.data_ptr(), t._version) for t in
                             itertools.chain(self.model.parameters(), self.model.buffers())))
        return (tuple(x.shape), x.dtype, x.device, ops.get_precision(), self.model.training, version)

    def __call__(self, x):
        if not x.is_cuda:
            return self.model(x)
        key = self._key(x)
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= self.max_graphs:     # (a new parameter version or shape: drop the oldest capture)
                self._graphs.pop(next(iter(self._graphs)))
            static_x = x.clone()
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):                        # warm-up: allocator, packed weights, derived filters
                    self.model(static_x)
            torch.cuda.current_stream(x.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph), torch.no_grad():
                static_y = self.model(static_x)
            ent = self._graphs[key] = (graph, static_x, static_y)
        graph, static_x, static_y = ent
        static_x.copy_(x)
        graph.replay()
        return static_y.clone() if self.copy_output else static_y


class PatchPredict:
    """Sliding-window prediction with overlap averaging (reference :105-152).

    padding_mode (reference :114,132 -> torchio GridSampler): None, a number (constant fill) or one of numpy.pad's
    'constant' / 'edge' / 'reflect' / 'symmetric' / 'wrap': the volume is treated as padded by patch_overlap // 2
    voxels per side before tiling, and the aggregate is cropped back (index math in the gather / finalize
    kernels; no padded copy).  The other numpy.pad modes (statistics, ramps) have no kernel.

    Multi-GPU (inside `distributed.unit_sharding()`): tile i -> rank i % world, one collective returns the tile
    outputs.  `result_on="all"` (default): all_gather, every rank aggregates and returns the volume;
    `result_on="rank0"`: gather to rank 0 only, which alone aggregates (the other ranks return None) -- no rank
    receives or re-aggregates tiles it does not need.  `timings` (a dict) accumulates seconds per phase
    (tile_gather, model, exchange, accumulate, finalize; a device synchronisation per stamp, for profiling).
    `graph=True`: full tile batches go through a `GraphedForward` of the model (launch-bound small patches).
    """

    def __init__(self, image_names: Sequence[str] = ("X",), patch_batch_size: int = 16, patch_size=None,
                 patch_overlap=(0, 0, 0), padding_mode=None, overlap_mode: str = "average", ops_backend=ops,
                 result_on: str = "all", timings: Optional[dict] = None, graph: bool = False):
        if overlap_mode != "average":
            raise NotImplementedError("only overlap_mode='average' (the mode the reference uses) is implemented")
        self.pad_value = 0.0
        if padding_mode is not None and not isinstance(padding_mode, str):
            self.pad_value, padding_mode = float(padding_mode), "constant"
        if padding_mode is not None and padding_mode not in ops.PAD_MODES:
            raise NotImplementedError(f"padding_mode={padding_mode!r} has no kernel (supported: None, a number, "
                                      f"{sorted(ops.PAD_MODES)})")
        if result_on not in ("all", "rank0"):
            raise ValueError("result_on must be 'all' or 'rank0'")
        self.image_names = image_names
        self.patch_batch_size = patch_batch_size
        self.patch_size = _triple(patch_size)
        self.patch_overlap = _triple(patch_overlap)
        self.padding_mode = padding_mode
        self.overlap_mode = overlap_mode
        self.result_on = result_on
        self.timings = timings
        self.graph = graph
        self._graphed = None     # (model, GraphedForward)
        self._ops = ops_backend  # the HIP ops; tests inject a CPU double for the gloo plumbing test

    def _stamp(self, name, t0, device):
        if self.timings is None:
            return t0
        import time
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        now = time.perf_counter()
        self.timings[name] = self.timings.get(name, 0.0) + (now - t0)
        return now

    def predict_volume(self, model, volume: torch.Tensor) -> Optional[torch.Tensor]:
        """volume [C, V0, V1, V2] on the device -> averaged prediction [C_out, V0, V1, V2]."""
        import time
        k = self._ops
        vshape = tuple(volume.shape[1:])
        border = tuple(o // 2 for o in self.patch_overlap) if self.padding_mode is not None else (0, 0, 0)
        pshape = tuple(v + 2 * b for v, b in zip(vshape, border))        # the (virtually) padded volume
        locs = grid_locations(pshape, self.patch_size, self.patch_overlap)
        dev = volume.device
        run = model
        if self.graph and dev.type == "cuda":
            if self._graphed is None or self._graphed[0] is not model:
                self._graphed = (model, GraphedForward(model, copy_output=False))
            run = self._graphed[1]
        t = time.perf_counter()
        with D.shard_scope() as sharded:  # tiles over ranks only inside distributed.unit_sharding(), outermost sharder
            world = torch.distributed.get_world_size() if sharded else 1
            rank = torch.distributed.get_rank() if sharded else 0
            mine = D.shard_indices(len(locs), rank, world)
            outs = []
            with torch.no_grad():
                for s in range(0, len(mine), self.patch_batch_size):
                    idx = mine[s:s + self.patch_batch_size]
                    loc = torch.tensor([locs[i] for i in idx], dtype=torch.int32, device=dev)
                    if self.padding_mode is None:
                        tiles_in = k.patch_gather(volume, loc, self.patch_size)
                    else:
                        tiles_in = k.patch_gather_padded(volume, loc, self.patch_size, border, self.padding_mode,
                                                         self.pad_value)
                    t = self._stamp("tile_gather", t, dev)
                    # (the graph's output buffer is reused by the next replay: torch.cat below copies, but only after
                    # the loop, so a graphed batch is cloned here; the ragged last batch is a second capture, keyed by
                    # its shape -- it recurs once per volume)
                    outs.append(run(tiles_in).clone() if run is not model else model(tiles_in))
                    t = self._stamp("model", t, dev)
        local = torch.cat(outs, dim=0) if outs else None
        meta = torch.tensor([local.shape[1] if local is not None else 0], device=dev)
        if world > 1:  # ranks without tiles learn the channel count (tiny, once per volume)
            torch.distributed.all_reduce(meta, op=torch.distributed.ReduceOp.MAX)
        c_out = int(meta.item())
        to_rank0 = world > 1 and self.result_on == "rank0"
        tiles = D.gather_tiles(local, len(locs), (c_out,) + self.patch_size, torch.float32, dev, sharded=world > 1,
                               dst=0 if to_rank0 else None)
        t = self._stamp("exchange", t, dev)
        if tiles is None:       # result_on="rank0" and this is another rank
            return None
        # aggregation in grid order: identical bits regardless of world size.  One pass over the tile grid (per voxel:
        # the covering tiles summed in tile order, divided by their count) where the backend has it -- no accumulator /
        # count volumes, no zero-fill, no read-modify-write; the same bits as the batch-by-batch loop below.
        if hasattr(k, "patch_aggregate_grid"):
            out = k.patch_aggregate_grid(tiles, grid_axes(pshape, self.patch_size, self.patch_overlap), vshape, border)
            self._stamp("aggregate", t, dev)
            return out
        accum = torch.zeros((c_out,) + pshape, dtype=torch.float32, device=dev)
        count = torch.zeros(pshape, dtype=torch.float32, device=dev)
        all_loc = torch.tensor(locs, dtype=torch.int32, device=dev)
        for s in range(0, len(locs), self.patch_batch_size):  # same batching as the reference loop (:136-141)
            k.patch_accumulate(tiles[s:s + self.patch_batch_size], all_loc[s:s + self.patch_batch_size], accum, count)
        t = self._stamp("accumulate", t, dev)
        out = k.patch_finalize(accum, count) if self.padding_mode is None else k.patch_finalize_crop(accum, count, border)
        self._stamp("finalize", t, dev)
        return out

    def predict(self, model, device, batch, label_attributes=None):
        x = batch["X"].to(device)
        batch = dict(batch)
        preds = [self.predict_volume(model, v) for v in x]
        batch["y_pred"] = None if any(p is None for p in preds) else torch.stack(preds)
        return batch
