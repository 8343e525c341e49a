"""Device-side patch sampling for patch training.

The reference feeds patch training through `tio.Queue` + `tio.WeightedSampler(patch_size,
probability_map='patch_probability')` in CPU worker processes (data_loader_factory.py:36-54,
research/msseg2/msseg2.py:148-149; the probability map comes from ImageFromLabels,
transforms/image_from_labels.py:11-57: e.g. brain = 1, lesion = 100).  At tens of patches per
second per GPU that loader starves the device, so here the volume stays resident in HBM and
patches are cut out by the `patch_gather` kernel at corner locations drawn on the device:

* UniformSampler  -- every corner that keeps the patch inside the volume is equally likely
  (tio.UniformSampler semantics);
* WeightedSampler -- the patch CENTRE is drawn with probability proportional to the map, restricted
  to centres whose patch fits in the volume (tio.WeightedSampler semantics: the border of the
  map is zeroed, the map is normalised, a voxel index is drawn from the cumulative distribution).

torchio is not available offline, so both follow its documented behaviour (parity unpinned, as for
the grid sampler); the tests check the distributional properties and the index arithmetic.
"""
from typing import Callable, Iterable, Iterator, Optional, Sequence, Tuple

import torch

from . import ops


def _triple(v):
    return tuple(int(a) for a in v) if isinstance(v, (tuple, list)) else (int(v),) * 3


class UniformSampler:
    def __init__(self, patch_size, ops_backend=ops):
        self.patch_size = _triple(patch_size)
        self._ops = ops_backend

    def sample_locations(self, volume_shape: Sequence[int], n: int, device, generator=None) -> torch.Tensor:
        """int32 [n, 3] corner indices (i0, j0, k0)."""
        hi = [s - p + 1 for s, p in zip(volume_shape, self.patch_size)]
        if min(hi) < 1:
            raise ValueError(f"patch size {self.patch_size} exceeds volume shape {tuple(volume_shape)}")
        cols = [torch.randint(0, h, (n,), device=device, generator=generator) for h in hi]
        return torch.stack(cols, dim=1).to(torch.int32)

    def __call__(self, volume: torch.Tensor, n: int, generator=None, extra: Optional[Sequence[torch.Tensor]] = None):
        """volume [C, V0, V1, V2] -> (patches [n, C, *patch], locations); `extra` volumes (labels,
        probability maps ...) are cut at the same locations."""
        loc = self.sample_locations(volume.shape[1:], n, volume.device, generator)
        out = [self._ops.patch_gather(volume, loc, self.patch_size)]
        for v in extra or ():
            out.append(self._ops.patch_gather(v, loc, self.patch_size))
        return (out[0] if not extra else tuple(out)), loc


class WeightedSampler(UniformSampler):
    def __init__(self, patch_size, ops_backend=ops):
        super().__init__(patch_size, ops_backend)

    def centre_distribution(self, probability_map: torch.Tensor) -> torch.Tensor:
        """Probability of each voxel being drawn as a patch centre: the map with every centre whose
        patch would leave the volume zeroed, normalised to 1 (flattened)."""
        pm = probability_map.reshape(probability_map.shape[-3:]).to(torch.float64).clamp_min(0)
        lo = [p // 2 for p in self.patch_size]                      # centre index inside the patch
        hi = [p - p // 2 - 1 for p in self.patch_size]              # voxels after the centre
        valid = torch.zeros_like(pm)
        sl = tuple(slice(l, s - h) for l, h, s in zip(lo, hi, pm.shape))
        valid[sl] = pm[sl]
        total = valid.sum()
        if not torch.isfinite(total) or total <= 0:
            raise RuntimeError("probability map has no positive entry where a patch fits")
        return (valid / total).flatten()

    def sample_locations(self, probability_map: torch.Tensor, n: int, generator=None) -> torch.Tensor:
        shape = probability_map.shape[-3:]
        pdf = self.centre_distribution(probability_map)
        cdf = torch.cumsum(pdf, dim=0)
        cdf = cdf / cdf[-1]
        u = torch.rand(n, dtype=torch.float64, device=cdf.device, generator=generator)
        # a draw at / beyond the last cdf value (rounding) falls back to the last centre with non-zero
        # probability -- never to the zeroed border, whose patch would leave the volume
        last_valid = torch.nonzero(pdf > 0)[-1, 0]
        flat = torch.minimum(torch.searchsorted(cdf, u, right=True), last_valid)
        k = flat % shape[2]
        j = (flat // shape[2]) % shape[1]
        i = flat // (shape[1] * shape[2])
        centre = torch.stack([i, j, k], dim=1)
        corner = centre - torch.tensor([p // 2 for p in self.patch_size], device=centre.device)
        hi = torch.tensor([s - p for s, p in zip(shape, self.patch_size)], device=centre.device)
        corner = torch.minimum(corner.clamp_min(0), hi)  # the gather kernel does no bounds checks of its own
        return corner.to(torch.int32)

    def __call__(self, volume: torch.Tensor, probability_map: torch.Tensor, n: int, generator=None,
                 extra: Optional[Sequence[torch.Tensor]] = None):
        loc = self.sample_locations(probability_map.to(volume.device), n, generator)
        out = [self._ops.patch_gather(volume, loc, self.patch_size)]
        for v in extra or ():
            out.append(self._ops.patch_gather(v, loc, self.patch_size))
        return (out[0] if not extra else tuple(out)), loc



class VolumeFeeder:
    """Double-buffered host -> device feeding of whole volumes for patch training (the feeding half of N2).

    The reference's patch loader (`tio.Queue`, data_loader_factory.py:36-54) cuts patches on the CPU and ships
    every patch batch over PCIe.  Here a SUBJECT's volumes cross the link once: while the samplers above cut
    patches from the resident volume(s) on the device, the next subject is staged into page-locked host memory and
    copied to the device on a side HIP stream (`non_blocking`), so the H2D transfer (a 4 x 256^3 fp32 volume is
    268 MB = ~4.5 ms at PCIe Gen5 x16) overlaps `patches_per_volume` training steps.  Two device slots and two
    pinned staging buffers per tensor name are reused for the whole run (no allocation in the loop).

        feeder = VolumeFeeder(subjects, device)             # subjects: iterable of {name: CPU tensor}
        for vols in feeder:                                  # vols: {name: device tensor}, valid until the next step
            for _ in range(patches_per_volume):
                (x, y), loc = sampler(vols["X"], vols["prob"], n, extra=[vols["y"]])
                ...
    """

    def __init__(self, subjects: Iterable[dict], device, pin_memory: bool = True):
        self.subjects = subjects
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.pin = pin_memory and self.cuda
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._slots = [{}, {}]       # device tensors per slot
        self._stage = [{}, {}]       # pinned host tensors per slot
        self._ready = [None, None]   # event: the slot's upload has finished
        self._free = [None, None]    # event: the consumer is done with the slot

    def _upload(self, slot: int, subject: dict):
        """stage `subject` into slot `slot` (asynchronously on the copy stream)"""
        dev, st = self._slots[slot], self._stage[slot]
        if self.cuda:
            if self._free[slot] is not None:
                self.copy_stream.wait_event(self._free[slot])   # the trainer has finished reading this slot
            ctx = torch.cuda.stream(self.copy_stream)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            for name, t in subject.items():
                if not torch.is_tensor(t):
                    dev[name] = t
                    continue
                if name not in dev or dev[name].shape != t.shape or dev[name].dtype != t.dtype:
                    dev[name] = torch.empty(t.shape, dtype=t.dtype, device=self.device)
                    if self.pin:
                        st[name] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                if self.pin:
                    st[name].copy_(t)                       # pageable -> pinned (host memcpy)
                    dev[name].copy_(st[name], non_blocking=True)
                else:
                    dev[name].copy_(t)
            for name in list(dev):
                if name not in subject:
                    del dev[name]
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
                self._ready[slot] = ev

    def __iter__(self) -> Iterator[dict]:
        it = iter(self.subjects)
        try:
            nxt = next(it)
        except StopIteration:
            return
        slot = 0
        self._upload(slot, nxt)
        while True:
            try:
                nxt = next(it)
                have_next = True
            except StopIteration:
                have_next = False
            if have_next:
                self._upload(slot ^ 1, nxt)                 # next subject in flight while this one is consumed
            if self.cuda:
                torch.cuda.current_stream(self.device).wait_event(self._ready[slot])
            yield dict(self._slots[slot])
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self._free[slot] = ev
            if not have_next:
                return
            slot ^= 1
