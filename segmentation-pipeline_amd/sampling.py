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
from typing import Optional, Sequence, Tuple

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
