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
    """tio.WeightedSampler semantics on the device.  The cumulative structure of a probability map is built ONCE per map
    (`m355_sampler_build`: a two-level table, 8 bytes per 1024 voxels -- cached per (storage, version, shape) of the
    map tensor); a batch of draws is one uniform-random launch plus ONE `m355_sampler_draw` launch (a binary search over
    the table and a scan of one 1024-voxel block per patch), then the usual gathers.  Nothing is synchronised with
    the host after the first use of a map (whose total weight is checked once)."""

    def __init__(self, patch_size, ops_backend=ops, max_cached_maps: int = 8):
        super().__init__(patch_size, ops_backend)
        self._tables = {}
        self._max = max_cached_maps

    def centre_distribution(self, probability_map: torch.Tensor) -> torch.Tensor:
        """Probability of each voxel being drawn as a patch centre: the map with every centre whose
        patch would leave the volume zeroed, normalised to 1 (flattened).  Host-side restatement (float64 torch ops) of
        what the table encodes; the sampling path itself does not use it."""
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

    def _table(self, pm: torch.Tensor):
        key = (pm.data_ptr(), pm._version, tuple(pm.shape), pm.device)
        ent = self._tables.get(key)
        if ent is None:
            table = self._ops.sampler_build(pm, self.patch_size)
            total = float(table[-1])                  # the one host synchronisation per map
            if not (total > 0) or total == float("inf"):
                raise RuntimeError("probability map has no positive entry where a patch fits")
            if len(self._tables) >= self._max:
                self._tables.pop(next(iter(self._tables)))
            ent = self._tables[key] = (table, pm)     # (keeps the map alive: the key holds its address)
        return ent[0]

    def sample_locations(self, probability_map: torch.Tensor, n: int, generator=None) -> torch.Tensor:
        pm = probability_map.reshape(probability_map.shape[-3:])
        if pm.dtype != torch.float32 or not pm.is_contiguous():
            pm = pm.float().contiguous()
        if any(p > s for p, s in zip(self.patch_size, pm.shape)):
            raise ValueError(f"patch size {self.patch_size} exceeds volume shape {tuple(pm.shape)}")
        u = torch.rand(n, dtype=torch.float64, device=pm.device, generator=generator)
        return self._ops.sampler_draw(pm, self._table(pm), self.patch_size, u)

    def __call__(self, volume: torch.Tensor, probability_map: torch.Tensor, n: int, generator=None,
                 extra: Optional[Sequence[torch.Tensor]] = None):
        if probability_map.device != volume.device:
            probability_map = probability_map.to(volume.device)
        loc = self.sample_locations(probability_map, n, generator)
        out = [self._ops.patch_gather(volume, loc, self.patch_size)]
        for v in extra or ():
            out.append(self._ops.patch_gather(v, loc, self.patch_size))
        return (out[0] if not extra else tuple(out)), loc



class VolumeFeeder:
    """Double-buffered host -> device feeding of whole volumes for patch training (the feeding half of N2).

    The reference's patch loader (`tio.Queue`, data_loader_factory.py:36-54) cuts patches on the CPU and ships
    every patch batch over PCIe.  Here a SUBJECT's volumes cross the link once: while the samplers above cut
    patches from the resident volume(s) on the device, a WORKER THREAD pulls the next subject from the iterable
    (whatever loading / decoding that does), stages it into page-locked host memory and enqueues the copy to the
    device on a side HIP stream, so neither the pageable -> pinned memcpy (a 4 x 256^3 fp32 volume is 268 MB: ~30 ms of
    host memcpy) nor the H2D transfer (~4.5 ms at PCIe Gen5 x16) ever runs on the thread that launches the training
    kernels.  Two device slots and two pinned staging buffers per tensor name are reused for the whole run.

    Ordering (per slot): the worker overwrites a slot's pinned buffer only after the HOST has seen the previous
    upload out of that buffer complete (`Event.synchronize`), and the upload into the slot's device tensors waits on
    the stream for the event the consumer recorded when it was done reading them; the consumer's stream waits for
    the upload event before the volumes are used.  The worker is at most one subject ahead.

        feeder = VolumeFeeder(subjects, device)             # subjects: iterable of {name: CPU tensor}
        for vols in feeder:                                  # vols: {name: device tensor}, valid until the next step
            for _ in range(patches_per_volume):
                (x, y), loc = sampler(vols["X"], vols["prob"], n, extra=[vols["y"]])
                ...
    """

    _STOP = object()

    def __init__(self, subjects: Iterable[dict], device, pin_memory: bool = True):
        self.subjects = subjects
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        if self.cuda and self.device.index is None:      # the worker thread needs the concrete device
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.pin = pin_memory and self.cuda
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._slots = [{}, {}]       # device tensors per slot
        self._stage = [{}, {}]       # pinned host tensors per slot
        self._ready = [None, None]   # event: the slot's upload has finished
        self.host_wait_s = 0.0       # time the CONSUMER thread spent waiting for the worker (0 when feeding keeps up)

    # -- event / stream primitives (overridden by the tests' doubles) ------------------------------------------
    def _new_event(self):
        return torch.cuda.Event() if self.cuda else None

    def _stage_copy(self, dst, src):
        dst.copy_(src)                                      # pageable -> pinned (host memcpy, worker thread)

    def _upload(self, slot: int, subject: dict, free_event):
        """stage `subject` into slot `slot` (worker thread; asynchronously on the copy stream)"""
        dev, st = self._slots[slot], self._stage[slot]
        if self.cuda:
            torch.cuda.set_device(self.device)
            if self._ready[slot] is not None:
                # the previous H2D copy OUT of this slot's pinned buffers must have finished before the host
                # overwrites them (a stream-side wait is not enough: the memcpy below runs now, on the host)
                self._ready[slot].synchronize()
            if free_event is not None:
                self.copy_stream.wait_event(free_event)     # the trainer has finished reading this slot's tensors
            ctx = torch.cuda.stream(self.copy_stream)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            for name, t in subject.items():
                if not torch.is_tensor(t):
                    dev[name] = t
                    continue
                if name not in dev or not torch.is_tensor(dev[name]) or dev[name].shape != t.shape or dev[name].dtype != t.dtype:
                    dev[name] = torch.empty(t.shape, dtype=t.dtype, device=self.device)
                    if self.pin:
                        st[name] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                if self.pin:
                    self._stage_copy(st[name], t)
                    dev[name].copy_(st[name], non_blocking=True)
                else:
                    dev[name].copy_(t)
            for name in list(dev):
                if name not in subject:
                    del dev[name]
            if self.cuda:
                ev = self._new_event()
                ev.record(self.copy_stream)
                self._ready[slot] = ev

    def _worker(self, ready_q, free_q, stop):
        try:
            for k, subject in enumerate(self.subjects):
                slot = k & 1
                free_event = None
                if k >= 2:                                   # slot reuse: wait (host) until the consumer left it
                    free_event = free_q[slot].get()
                    if free_event is self._STOP:
                        return
                if stop.is_set():
                    return
                self._upload(slot, subject, free_event)
                ready_q.put((slot, self._ready[slot], None))
            ready_q.put((None, None, None))
        except BaseException as exc:   # noqa: BLE001 -- re-raised on the consumer thread
            ready_q.put((None, None, exc))

    def __iter__(self) -> Iterator[dict]:
        import queue
        import threading
        import time
        ready_q = queue.Queue()
        free_q = [queue.Queue(), queue.Queue()]
        stop = threading.Event()
        worker = threading.Thread(target=self._worker, args=(ready_q, free_q, stop), daemon=True, name="m355-volume-feeder")
        worker.start()
        try:
            while True:
                t0 = time.perf_counter()
                slot, ready, exc = ready_q.get()
                self.host_wait_s += time.perf_counter() - t0
                if exc is not None:
                    raise exc
                if slot is None:
                    return
                if self.cuda:
                    torch.cuda.current_stream(self.device).wait_event(ready)
                yield dict(self._slots[slot])
                ev = None
                if self.cuda:
                    ev = self._new_event()
                    ev.record(torch.cuda.current_stream(self.device))
                free_q[slot].put(ev)
        finally:
            stop.set()
            for q in free_q:
                q.put(self._STOP)
            worker.join(timeout=60)
