"""Drop-in models on the GPU vs golden vectors produced by the real reference modules.

Tolerances (north star): probabilities and Dice within 1e-4 absolute in fp32; argmax
masks bit-exact on the structured volume; gradients within 1e-3 relative to the
tensor's max (different summation order through ~20 conv layers).
"""
from functools import partial

import numpy as np
import pytest
import torch
from torch import nn

from segmentation_pipeline_amd import ops
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
from segmentation_pipeline_amd.models import (BlurConv3d, BlurConvTranspose3d, EnsembleFlips, EnsembleModels,
                                              EnsembleOrientations, ModularUNet, NestedResUNet, StochasticMatrix,
                                              WSConv3d)

pytestmark = pytest.mark.gpu

GN8 = {'normalization_class': partial(nn.GroupNorm, 8)}
CONVT = dict(upsample_class=nn.ConvTranspose3d, upsample_params={'kernel_size': 2, 'stride': 2})
PROB_TOL = 1e-4

BUILDERS = {
    "unet_default_bn.npz": (lambda: ModularUNet(4, 3, [8, 16, 32], 3), None),
    "unet_gn_convt.npz": (lambda: ModularUNet(4, 3, [8, 16, 32], 3, block_params=dict(GN8), **CONVT), None),
    "unet_res_blur.npz": (lambda: ModularUNet(
        2, 2, [8, 8, 16], 3, block_params={'residual': True}, downsample_class=BlurConv3d,
        downsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1}, upsample_class=BlurConvTranspose3d,
        upsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1, 'output_padding': 0}), [1, 100]),
    "nested_res_unet.npz": (lambda: NestedResUNet(3, 2, 8), None),
}


def maxerr(a, b):
    return (a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max().item()


def grad_close(got, ref, name, rtol=1e-3):
    ref = torch.as_tensor(ref).double()
    err = (got.detach().cpu().double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= rtol * scale + 1e-7, f"grad {name}: err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("name", list(BUILDERS))
def test_model_matches_reference_golden(golden, name):
    g = golden(name)
    build, cw = BUILDERS[name]
    model = build()
    model.load_state_dict(g.state_dict("m.sd."))
    model = model.cuda()
    crit = HybridLogisticDiceLoss(logistic_class_weights=cw)
    x, y = g.t("x").cuda(), g.t("y").cuda()

    model.train()
    p = model(x)
    assert p.shape == tuple(g["m.probs_train"].shape)
    assert maxerr(p, g["m.probs_train"]) <= PROB_TOL
    ld = crit(p, y)
    assert set(ld) == {"loss", "dice_loss", "logistic_loss"}
    for k in ld:
        assert abs(ld[k].item() - float(g[f"m.{k}"])) <= 1e-4, k
    model.zero_grad()
    ld["loss"].backward()
    n_checked = 0
    for k, v in model.named_parameters():
        gk = f"m.grad.{k}"
        if gk in g.keys():
            assert v.grad is not None, k
            grad_close(v.grad, g[gk], k)
            n_checked += 1
        else:
            assert v.grad is None, f"{k} is unused in the reference (grad None) but got a gradient"
    assert n_checked > 0
    # BatchNorm running statistics moved exactly like torch's
    after = g.state_dict("m.sd_after.")
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            assert maxerr(v.float(), after[k].float()) <= 1e-5, k

    model.eval()
    with torch.no_grad():
        pe = model(x)
    assert maxerr(pe, g["m.probs_eval"]) <= PROB_TOL


def test_argmax_bit_exact_on_structured_volume(golden):
    g = golden("unet_gn_convt.npz")
    model = BUILDERS["unet_gn_convt.npz"][0]()
    model.load_state_dict(g.state_dict("struct.sd."))
    model = model.cuda().eval()
    with torch.no_grad():
        p = model(g.t("xs").cuda())
    assert maxerr(p, g["probs_struct"]) <= PROB_TOL
    assert (g["struct_class_hist"] > 0).all(), "every class must win somewhere"
    assert float(g["min_top2_gap"]) > 10 * PROB_TOL, "fixture must have clear winners"
    am, counts = ops.argmax_confusion(p, g.t("argmax_struct").cuda())
    assert torch.equal(am.cpu(), g.t("argmax_struct"))
    # confusion against itself: only TP / TN populated
    assert counts[..., 1].sum().item() == 0 and counts[..., 2].sum().item() == 0


def test_sgd_trajectory_matches_reference(golden):
    """segmentation_trainer.py:162-180 order with SGD(lr=1e-3, momentum=0.95) (msseg2.py:94)."""
    g = golden("unet_gn_convt.npz")
    model = BUILDERS["unet_gn_convt.npz"][0]()
    model.load_state_dict(g.state_dict("m.sd."))
    model = model.cuda()
    crit = HybridLogisticDiceLoss()
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.95)
    x, y = g.t("x").cuda(), g.t("y").cuda()
    losses = []
    for _ in range(3):
        model.train()
        ld = crit(model(x), y)
        opt.zero_grad()
        ld["loss"].backward()
        opt.step()
        model.eval()
        losses.append([ld["loss"].item(), ld["dice_loss"].item(), ld["logistic_loss"].item()])
    np.testing.assert_allclose(np.asarray(losses), g["sgd_losses"], rtol=0, atol=1e-4)
    final = g.state_dict("sgd.sd_final.")
    for k, v in model.state_dict().items():
        assert maxerr(v, final[k]) <= 1e-5, k


def _adam_update_error(model, g):
    """relative L2 distance between this model's total parameter update (final - initial) and the reference's, over all
    parameters jointly, and the worst BatchNorm running-statistics error"""
    sd0, sdf = g.state_dict("sd0."), g.state_dict("sd_final.")
    num = den = 0.0
    worst_buf = 0.0
    params = dict(model.named_parameters())
    for k, v in model.state_dict().items():
        if k in params:
            d_ref = (sdf[k].double() - sd0[k].double())
            d_gpu = (v.detach().cpu().double() - sd0[k].double())
            num += float(((d_gpu - d_ref) ** 2).sum())
            den += float((d_ref ** 2).sum())
        elif v.is_floating_point():
            worst_buf = max(worst_buf, maxerr(v, sdf[k]))
    return (num / den) ** 0.5, worst_buf


def _adam_batches(g, n=4):
    return [{"X": g.t(f"x{i}").cuda(), "y": g.t(f"y{i}").cuda()} for i in range(n)]


def test_adam_trajectory_matches_reference(golden):
    """The reference's own optimiser configuration end to end (research/dmri_hippo/configs/main_config.py:123-128:
    NestedResUNet + Adam(lr=2e-4) + HybridLogisticDiceLoss; loop order segmentation_trainer.py:162-180): 4 Adam steps on
    changing batches against the trajectory of the real reference modules on torch-CPU (tools/gen_golden.py::gen_round4).
    Adam's first step moves EVERY element by lr * sign(g) (m / sqrt(v) = g / |g|), so an element whose gradient is below
    the fp32 summation-order noise of its tensor (|g| < ~1e-4 of the tensor's scale: 1-2 % of the elements of a
    random-init network) moves by +-lr on the sign of that noise, here as on any other backend.  The update is therefore
    compared as a whole -- relative L2 over all parameters (measured 2.3e-2, the size of that sign population; a wrong
    gradient scale, a dropped term or a stale moment is O(1)) -- and the losses, moments and running statistics
    element by element."""
    g = golden("round4.npz")
    model = NestedResUNet(3, 2, 8)
    model.load_state_dict(g.state_dict("sd0."))
    model = model.cuda()
    crit = HybridLogisticDiceLoss()
    opt = torch.optim.Adam(model.parameters(), lr=2e-4)
    losses = []
    for b in _adam_batches(g):
        model.train()
        ld = crit(model(b["X"]), b["y"])
        opt.zero_grad()
        ld["loss"].backward()
        opt.step()
        model.eval()
        losses.append([ld["loss"].item(), ld["dice_loss"].item(), ld["logistic_loss"].item()])
    np.testing.assert_allclose(np.asarray(losses), g["adam_losses"], rtol=0, atol=1e-4)
    rel, worst_buf = _adam_update_error(model, g)
    print(f"adam eager: update rel-L2 {rel:.3e}, worst running-stat error {worst_buf:.3e}")
    assert rel <= 5e-2 and worst_buf <= 1e-4
    names = [str(k) for k in g["param_names"]]
    ea = np.asarray([opt.state[p]["exp_avg"].double().norm().item() for p in model.parameters()])
    es = np.asarray([opt.state[p]["exp_avg_sq"].double().norm().item() for p in model.parameters()])
    assert names == [k for k, _ in model.named_parameters()]
    # (moments after four steps of two trajectories that differ by the sign population: the small tensors -- norm affine
    # parameters, biases -- measured up to 6.6 % apart in norm, the conv weights ~0.1 %)
    np.testing.assert_allclose(ea, g["exp_avg_norms"], rtol=0.15, atol=1e-9)
    np.testing.assert_allclose(es, g["exp_avg_sq_norms"], rtol=0.3, atol=1e-12)
    big = g["exp_avg_norms"] > 5e-3
    np.testing.assert_allclose(ea[big], g["exp_avg_norms"][big], rtol=1e-2)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_adam_through_graphed_train_step(golden, mode):
    """GraphedTrainStep needs Adam(capturable=True) (the step counter lives on the device): the replayed trajectory is
    bit-identical to the eager loop with the same optimiser, and (fp32) follows the reference trajectory."""
    import copy
    import segmentation_pipeline_amd as sp
    from segmentation_pipeline_amd.trainer import GraphedTrainStep
    g = golden("round4.npz")
    m_e = NestedResUNet(3, 2, 8)
    m_e.load_state_dict(g.state_dict("sd0."))
    m_e = m_e.cuda()
    m_g = copy.deepcopy(m_e)
    crit = HybridLogisticDiceLoss()
    batches = _adam_batches(g)
    with sp.precision(mode):
        opt_e = torch.optim.Adam(m_e.parameters(), lr=2e-4, capturable=True)
        opt_g = torch.optim.Adam(m_g.parameters(), lr=2e-4, capturable=True)
        step = GraphedTrainStep(m_g, crit, opt_g, warmup=1)
        le, lg = [], []
        for b in batches:
            m_e.train()
            opt_e.zero_grad(set_to_none=True)
            ld = crit(m_e(b["X"]), b["y"])
            ld["loss"].backward()
            opt_e.step()
            le.append(ld["loss"].detach().clone())
            lg.append(step(b)["loss"])
    assert torch.equal(torch.stack(le), torch.stack(lg))
    for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a, b), k
    if mode == "fp32":
        np.testing.assert_allclose(torch.stack(lg).cpu().numpy(), g["adam_losses"][:, 0], rtol=0, atol=1e-4)
        rel, worst_buf = _adam_update_error(m_g, g)
        assert rel <= 5e-2 and worst_buf <= 1e-4


@pytest.mark.parametrize("mode", ["fp32", "fp32_mfma"])
def test_cfg2_architecture_reduced_patch(golden, mode):
    """The real 5-level [32,64,128,256,320] GN/ConvT network (18.08 M params) on a 32^3 patch;
    weights re-created from seed 0 exactly as the fixture generator did.  Both fp32 arithmetics against the SAME
    reference vectors at the same tolerances: "fp32" (the wide convolutions as six bf16 MFMAs on the exact three-way
    operand split) and "fp32_mfma" (every product on the fp32 MFMA)."""
    import segmentation_pipeline_amd as sp
    with sp.precision(mode):
        _cfg2_architecture_reduced_patch(golden)


def _cfg2_architecture_reduced_patch(golden):
    g = golden("cfg2_arch_32cube.npz")
    torch.manual_seed(0)
    model = ModularUNet(4, 3, [32, 64, 128, 256, 320], 5, block_params=dict(GN8), **CONVT)
    np.testing.assert_allclose([p.double().sum().item() for p in model.parameters()], g["param_sums"], rtol=1e-10, atol=1e-10)
    model = model.cuda().train()
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn((1, 4, 32, 32, 32), generator=gen)
    lab = torch.randint(0, 3, (1, 32, 32, 32), generator=gen)
    y = torch.nn.functional.one_hot(lab, 3).permute(0, 4, 1, 2, 3).float().contiguous()
    p = model(x.cuda())
    assert maxerr(p[:, :, ::3, ::3, ::3], g["probs_sub"]) <= PROB_TOL
    am = p.argmax(dim=1).cpu().numpy().astype(np.int8)
    clear = g["top2_gap"] > 10 * PROB_TOL
    assert clear.mean() > 0.9
    assert np.array_equal(am[clear], g["argmax"][clear]), "argmax differs where the reference has a clear winner"
    ld = HybridLogisticDiceLoss()(p, y.cuda())
    np.testing.assert_allclose([ld["loss"].item(), ld["dice_loss"].item(), ld["logistic_loss"].item()], g["losses"],
                               rtol=0, atol=1e-4)
    ld["loss"].backward()
    norms = np.asarray([q.grad.double().norm().item() for q in model.parameters()])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-9)
    heads = np.stack([np.resize(q.grad.flatten()[:8].cpu().numpy(), 8) for q in model.parameters()])
    scale = np.abs(g["grad_heads"]).max(axis=1, keepdims=True) + 1e-12
    assert (np.abs(heads - g["grad_heads"]) / scale).max() <= 5e-2


def test_blur_ws_stochastic_components(golden):
    g = golden("components.npz")
    bc = BlurConv3d(8, 8, 3, stride=2, padding=1)
    bc.load_state_dict(g.state_dict("blur.sd."))
    bc = bc.cuda()
    x = g.t("blur.x").cuda().requires_grad_()
    y = bc(x)
    assert maxerr(y, g["blur.y"]) <= 1e-5
    y.sum().backward()
    grad_close(x.grad, g["blur.dx"], "blur.dx")
    grad_close(bc.weight.grad, g["blur.dw"], "blur.dw")
    assert bc.bias.grad is None  # unused parameter, as in the reference

    bt = BlurConvTranspose3d(8, 8, 3, stride=2, padding=1, output_padding=0, weight_standardization=True)
    bt.load_state_dict(g.state_dict("blurT.sd."))
    bt = bt.cuda()
    x = g.t("blurT.x").cuda().requires_grad_()
    y = bt(x)
    assert maxerr(y, g["blurT.y"]) <= 1e-4
    (y * y).sum().backward()
    grad_close(x.grad, g["blurT.dx"], "blurT.dx")
    grad_close(bt.weight.grad, g["blurT.dw"], "blurT.dw", rtol=2e-3)

    ws = WSConv3d(4, 6, 3, padding=1)
    ws.load_state_dict(g.state_dict("ws.sd."))
    assert maxerr(ws.cuda()(g.t("ws.x").cuda()), g["ws.y"]) <= 1e-4

    sm = StochasticMatrix(2, diag_bias=5)
    assert maxerr(sm(g.t("sm.x").cuda()), g["sm.y"]) <= 1e-6
    assert maxerr(sm(torch.zeros(1, 4, 1, 1, 1, device="cuda")), g["sm.zeros"]) <= 1e-6


def test_ensemble_flips_golden(golden):
    g = golden("components.npz")
    model = ModularUNet(4, 3, [8, 16], 2, block_params=dict(GN8), **CONVT)
    model.load_state_dict(g.state_dict("flips.sd."))
    model = model.cuda().eval()
    x = g.t("flips.x").cuda()
    with torch.no_grad():
        assert maxerr(EnsembleFlips(model, "mean")(x), g["flips.mean"]) <= PROB_TOL
        maj = EnsembleFlips(model, "majority", spatial_dims=(3, 4))(x)
    assert torch.equal(maj.cpu(), g.t("flips.majority34"))


def _ens_members(g):
    members = []
    for i in (0, 1):
        m = ModularUNet(2, 3, [8, 16], 2, block_params=dict(GN8), **CONVT)
        m.load_state_dict(g.state_dict(f"m{i}.sd."))
        members.append(m.cuda().eval())
    return members


def test_ensemble_orientations_models_and_nested_mean_golden(golden):
    """models/ensemble.py:38-103 on two tiny members: 48 orientations, model ensembles, and the ensemble of
    flip-ensembles; 'mean' within the fp32 tolerance.  ('majority' masks: the decisive-vote fixture below.)"""
    g = golden("ensembles_ws.npz")
    members = _ens_members(g)
    x = g.t("x").cuda()
    with torch.no_grad():
        assert maxerr(EnsembleOrientations(members[0], "mean")(x), g["orient.mean"]) <= PROB_TOL
        assert maxerr(EnsembleModels(members, "mean")(x), g["models.mean"]) <= PROB_TOL
        nested = EnsembleModels([EnsembleFlips(m, "mean", spatial_dims=(3, 4)) for m in members], "mean")
        assert maxerr(nested(x), g["nested.mean"]) <= PROB_TOL


def _decisive_members(g):
    members = []
    for i in (0, 1):
        m = ModularUNet(2, 3, [8, 16], 2, block_params=dict(GN8), **CONVT)
        m.load_state_dict(g.state_dict(f"ens.m{i}.sd."))
        members.append(m.cuda().eval())
    return members


def _majority_cases(members):
    return {
        "orient": EnsembleOrientations(members[0], "majority"),
        "flips": EnsembleFlips(members[1], "majority"),
        "models": EnsembleModels(members, "majority"),
        # the reference's production inference: majority of majorities (ms-inference.py:115-125)
        "nested_flips": EnsembleModels([EnsembleFlips(m, "majority") for m in members], "majority"),
        "nested_orient": EnsembleModels([EnsembleOrientations(m, "majority") for m in members], "majority"),
    }


def test_majority_ensembles_bit_exact_on_decisive_votes(golden):
    """'majority' = argmax per member -> mode over members (ties between vote COUNTS: smallest class, torch.mode on the
    CPU) -> int64 one-hot: integer work.  The fixture (tools/gen_golden.gen_round3, from the real reference) is built so
    that every member prediction has a top-2 probability gap >= 4e-4 > 2 x PROB_TOL at every voxel -- no admissible
    fp32 error can move a member's argmax -- so every mask must be reproduced BIT-EXACTLY, including the vote-count
    ties of the 2-member and nested ensembles."""
    g = golden("round3.npz")
    assert float(g["ens.min_top2_gap"]) >= 4e-4
    members = _decisive_members(g)
    x = g.t("ens.x").cuda()
    with torch.no_grad():
        for name, ens in _majority_cases(members).items():
            got = ens(x)
            assert got.dtype == torch.int64
            assert torch.equal(got.cpu(), g.t(f"ens.{name}.majority")), f"{name}: one-hot mask differs from the reference"
        # the decisiveness claim itself, measured on the GPU members
        for m in members:
            t = m(x).topk(2, dim=1).values
            assert (t[:, 0] - t[:, 1]).min().item() >= 4e-4 - PROB_TOL


def test_majority_tie_rule_smallest_class_wins():
    """vote-count ties resolve to the smallest class (torch.mode on CPU tensors, models/ensemble.py:29-31)"""
    from segmentation_pipeline_amd.models.ensemble import apply_strategy

    class Const(nn.Module):
        def __init__(self, cls):
            super().__init__()
            self.cls = cls

        def forward(self, x):
            p = torch.full((x.shape[0], 4) + tuple(x.shape[2:]), 0.1, device=x.device)
            p[:, self.cls] = 0.7
            return p
    x = torch.zeros((1, 1, 4, 4, 4), device="cuda")
    for classes in ([3, 1], [2, 0, 2, 0], [3, 3, 1, 1, 2]):
        got = EnsembleModels([Const(c) for c in classes], "majority")(x)
        ref = apply_strategy([Const(c)(x.cpu()) for c in classes], "majority")
        assert torch.equal(got.cpu(), ref)
        assert got[0, :, 0, 0, 0].argmax().item() == min(c for c in set(classes) if classes.count(c) == max(map(classes.count, classes)))


def test_cascade_configuration_golden(golden):
    """research/dmri_hippo/configs/cascade.py:53-66,75-78 as a model: ModularUNet(3, 16, [40, 80, 120], 3, residual
    blocks, Blur down / up, hypothesis StochasticMatrix(4, diag_bias=5)) under StandardPredict(sagittal_split=True);
    train-mode prediction, a scalar loss, every parameter gradient (norms + heads), running statistics, eval-mode
    prediction -- against the real reference (weights re-created from seed 0; init equality checked first)."""
    from segmentation_pipeline_amd.prediction import StandardPredict
    g = golden("round3.npz")
    torch.manual_seed(0)
    model = ModularUNet(3, 16, [40, 80, 120], 3, block_params={'residual': True}, downsample_class=BlurConv3d,
                        downsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1}, upsample_class=BlurConvTranspose3d,
                        upsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1, 'output_padding': 0},
                        hypothesis_class=StochasticMatrix, hypothesis_params={"channels": 4, "diag_bias": 5})
    assert sum(p.numel() for p in model.parameters()) == int(g["cascade.n_params"])
    np.testing.assert_allclose([p.double().sum().item() for p in model.parameters()], g["cascade.param_sums"], rtol=0, atol=1e-9)
    np.testing.assert_allclose([p.double().abs().sum().item() for p in model.parameters()], g["cascade.param_abs_sums"], rtol=1e-12)
    model = model.cuda().train()
    predictor = StandardPredict(sagittal_split=True, image_names=['X', 'y'], refine_image="y_prior")
    batch = predictor.predict(model, torch.device("cuda"), {"X": g.t("cascade.x")})
    y_pred = batch["y_pred"]
    assert maxerr(y_pred, g["cascade.y_pred"]) <= PROB_TOL
    cols = y_pred.reshape(1, 4, 4, -1).sum(dim=1)          # a column-stochastic matrix per voxel
    assert (cols - 1).abs().max().item() <= 1e-5
    w = g.t("cascade.w").cuda()
    loss = (y_pred * w).sum() / w.numel()
    assert abs(loss.item() - float(g["cascade.loss"])) <= 1e-6
    loss.backward()
    named = [(k, v) for k, v in model.named_parameters() if v.grad is not None]
    assert [k for k, _ in named] == list(g["cascade.grad_names"])       # same unused parameters (Blur biases) as the reference
    norms = np.asarray([v.grad.double().norm().item() for _, v in named])
    np.testing.assert_allclose(norms, g["cascade.grad_norms"], rtol=2e-3, atol=1e-10)
    heads = np.stack([np.resize(v.grad.flatten()[:8].cpu().numpy(), 8) for _, v in named])
    scale = np.abs(g["cascade.grad_heads"]).max(axis=1, keepdims=True) + 1e-12
    assert (np.abs(heads - g["cascade.grad_heads"]) / scale).max() <= 5e-2
    bufs = dict(model.named_buffers())
    rm = sum(v.double().sum().item() for k, v in bufs.items() if k.endswith("running_mean"))
    rv = sum(v.double().sum().item() for k, v in bufs.items() if k.endswith("running_var"))
    assert abs(rm - float(g["cascade.running_mean_sum"])) <= 1e-4 and abs(rv - float(g["cascade.running_var_sum"])) <= 1e-3 * abs(rv)
    model.eval()
    with torch.no_grad():
        pe = predictor.predict(model, torch.device("cuda"), {"X": g.t("cascade.x")})["y_pred"]
    assert maxerr(pe, g["cascade.y_pred_eval"]) <= PROB_TOL


def test_packed_weights_are_shared_across_input_shapes():
    """One packed buffer per (weight, direction, arithmetic) whatever the spatial extent / batch (the cache
    used to be keyed on the input shape and grew by a full copy of every weight per distinct validation volume size)."""
    torch.manual_seed(0)
    model = ModularUNet(4, 3, [8, 16], 2, block_params=dict(GN8), **CONVT).cuda().eval()
    ref = None
    shapes = [(1, 4, 8, 8, 8), (1, 4, 16, 8, 8), (2, 4, 8, 16, 24), (1, 4, 32, 32, 32), (1, 4, 8, 8, 40)]
    with torch.no_grad():
        outs = [model(torch.randn(s, generator=torch.Generator().manual_seed(1)).cuda()) for s in shapes]
    n_fwd = [len([k for k in p._m355_packed[2] if k[0] == 0]) for p in model.parameters() if hasattr(p, "_m355_packed")]
    assert n_fwd and all(n <= 2 for n in n_fwd), n_fwd       # (the Cout <= 4 out conv has two kernel families by extent)
    conv_w = model.down_blocks[0].layers.conv1.weight
    assert len(conv_w._m355_packed[2]) == 1
    # the shared buffer produces the same result as a model that only ever saw that shape
    torch.manual_seed(0)
    fresh = ModularUNet(4, 3, [8, 16], 2, block_params=dict(GN8), **CONVT).cuda().eval()
    with torch.no_grad():
        again = fresh(torch.randn(shapes[2], generator=torch.Generator().manual_seed(1)).cuda())
    assert torch.equal(again, outs[2])
    # training: forward + data-gradient forms, still one each, re-packed by ONE batched launch after the step
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for s in shapes[:3]:
        opt.zero_grad()
        model(torch.randn(s, generator=torch.Generator().manual_seed(2)).cuda()).square().mean().backward()
        opt.step()
    assert sorted(k[0] for k in conv_w._m355_packed[2]) == [0, 1]


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_graphed_train_step_redraws_dropout_masks(mode):
    """Dropout3d inside a captured train step (dmri_hippo: NestedResUNet(dropout_p=0.2)): the channel masks come from torch's
    CUDA generator, which moves on with every replay -- with lr = 0 the mask is the only thing that changes, and the
    losses of consecutive replays differ; eval() forwards outside the graph stay deterministic."""
    import segmentation_pipeline_amd as sp
    from segmentation_pipeline_amd.trainer import GraphedTrainStep
    torch.manual_seed(0)
    m = NestedResUNet(3, 2, 8, dropout_p=0.5).cuda()
    x = torch.randn(2, 3, 16, 16, 16, device="cuda")
    y = torch.nn.functional.one_hot(torch.randint(0, 2, (2, 16, 16, 16), device="cuda"), 2).permute(0, 4, 1, 2, 3).float().contiguous()
    with sp.precision(mode):
        step = GraphedTrainStep(m, HybridLogisticDiceLoss(), torch.optim.SGD(m.parameters(), lr=0.0))
        losses = [float(step({"X": x, "y": y})["loss"]) for _ in range(7)]
        assert all(l == l and abs(l) < 10 for l in losses)
        assert len(set(losses[1:])) >= 5, losses
        m.eval()
        with torch.no_grad():
            assert torch.equal(m(x), m(x))


@pytest.mark.parametrize("mode,norm", [("fp32", "group"), ("bf16", "group"), ("bf16", "batch")])
def test_graphed_train_step_reproduces_the_eager_trajectory(mode, norm):
    """trainer.GraphedTrainStep: the whole training iteration replayed from a hipGraph -- losses, final weights and
    BatchNorm running statistics bit-identical to the eager loop over 8 steps with changing batches (fp32, and the c8
    training flow of the 16-bit modes incl. the batched weight re-pack inside the captured step)."""
    import copy
    import segmentation_pipeline_amd as sp
    from segmentation_pipeline_amd.trainer import GraphedTrainStep
    torch.manual_seed(0)
    bp = dict(GN8) if norm == "group" else {}
    m_e = ModularUNet(4, 3, [8, 16], 2, block_params=bp, **CONVT).cuda().train()
    m_g = copy.deepcopy(m_e)
    g = torch.Generator().manual_seed(21)
    batches = []
    for _ in range(8):
        x = torch.randn((2, 4, 16, 16, 16), generator=g)
        lab = torch.randint(0, 3, (2, 16, 16, 16), generator=g)
        batches.append({"X": x.cuda(), "y": torch.nn.functional.one_hot(lab, 3).permute(0, 4, 1, 2, 3).float().contiguous().cuda()})
    crit = HybridLogisticDiceLoss()
    with sp.precision(mode):
        opt_e = torch.optim.SGD(m_e.parameters(), lr=1e-2, momentum=0.9)
        opt_g = torch.optim.SGD(m_g.parameters(), lr=1e-2, momentum=0.9)
        step = GraphedTrainStep(m_g, crit, opt_g, warmup=3)
        losses_e, losses_g = [], []
        # an UNMODIFIED eager loop: the stepper's first 3 calls are eager steps on their own batches, call 4 captures
        # and replays (the round-3 stepper applied batch 0 three times)
        for b in batches:
            opt_e.zero_grad(set_to_none=True)
            ld = crit(m_e(b["X"]), b["y"])
            ld["loss"].backward()
            opt_e.step()
            losses_e.append(ld["loss"].detach().clone())
            losses_g.append(step(b)["loss"])
    assert torch.equal(torch.stack(losses_e), torch.stack(losses_g))
    for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a, b), k
    assert len(step._graphs) == 1 and next(iter(step._graphs.values()))["graph"] is not None


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_two_streams_over_one_model_run_concurrently(tuning, mode):
    """the work-queue state of the queue-driven conv kernels used to live in the tail of the packed-
    weight buffer, so two launches over ONE model on different streams (validation overlapping training) corrupted each
    other's queues.  It now lives in a per-(device, stream) slot: forwards issued alternately on two streams -- the
    queue-driven kernels forced on these small shapes, few workgroups so that each walks many items -- reproduce the
    serial results bit for bit."""
    import segmentation_pipeline_amd as sp
    tuning(M355_CONV_PERSISTENT=2, M355_CONV_SLOTS=6, M355_H16_ONESHOT=3)
    torch.manual_seed(0)
    model = ModularUNet(4, 3, [16, 32], 2, block_params=dict(GN8), **CONVT).cuda().eval()
    g = torch.Generator().manual_seed(3)
    xa, xb = torch.randn((1, 4, 32, 32, 64), generator=g).cuda(), torch.randn((2, 4, 16, 32, 32), generator=g).cuda()
    with sp.precision(mode), torch.no_grad():
        ya, yb = model(xa), model(xb)          # serial reference (also packs the weights once)
        torch.cuda.synchronize()
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for _ in range(12):                    # one host thread alternates: the two streams' kernels overlap on the GPU
            with torch.cuda.stream(sa):
                oa = model(xa)
            with torch.cuda.stream(sb):
                ob = model(xb)
            outs.append((oa, ob))
        torch.cuda.synchronize()
    for oa, ob in outs:
        assert torch.equal(oa, ya) and torch.equal(ob, yb)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_two_captured_graphs_replayed_concurrently_do_not_share_queue_state(tuning, mode):
    """VERDICT r3 item 6 / ADVICE r3: torch captures every graph on one shared side stream, so keyed by stream alone a
    captured validation forward (prediction.GraphedForward) and a captured train step (trainer.GraphedTrainStep) baked in
    the SAME work-queue slot; replayed concurrently on two streams they would corrupt each other's tickets.  The slot is
    now keyed by (device, stream, capture id).  Queue-driven kernels forced, few workgroups (many items each); the
    validation model is a frozen copy so that the expected outputs do not depend on the interleaving."""
    import copy
    import segmentation_pipeline_amd as sp
    from segmentation_pipeline_amd.prediction import GraphedForward
    from segmentation_pipeline_amd.trainer import GraphedTrainStep
    tuning(M355_CONV_PERSISTENT=2, M355_CONV_SLOTS=6, M355_H16_ONESHOT=3)
    torch.manual_seed(0)
    m_t = ModularUNet(4, 3, [16, 32], 2, block_params=dict(GN8), **CONVT).cuda()
    m_s = copy.deepcopy(m_t)                     # serial twin of the training model
    m_v = copy.deepcopy(m_t).eval()              # the validation model (never updated)
    g = torch.Generator().manual_seed(5)
    xv = torch.randn((1, 4, 32, 32, 64), generator=g).cuda()
    batches = []
    for _ in range(10):
        x = torch.randn((2, 4, 16, 32, 32), generator=g)
        lab = torch.randint(0, 3, (2, 16, 32, 32), generator=g)
        batches.append({"X": x.cuda(), "y": torch.nn.functional.one_hot(lab, 3).permute(0, 4, 1, 2, 3).float().contiguous().cuda()})
    crit = HybridLogisticDiceLoss()
    with sp.precision(mode):
        # serial reference: plain eager training + eager validation forward
        opt_s = torch.optim.SGD(m_s.parameters(), lr=1e-2, momentum=0.9)
        losses_s = []
        for b in batches:
            m_s.train()
            opt_s.zero_grad(set_to_none=True)
            ld = crit(m_s(b["X"]), b["y"])
            ld["loss"].backward()
            opt_s.step()
            losses_s.append(ld["loss"].detach().clone())
        with torch.no_grad():
            yv = m_v(xv)
        torch.cuda.synchronize()
        # both captured (each capture runs on torch's shared capture stream), then replayed on two streams
        opt_t = torch.optim.SGD(m_t.parameters(), lr=1e-2, momentum=0.9)
        step = GraphedTrainStep(m_t, crit, opt_t, warmup=2)
        fwd = GraphedForward(m_v, copy_output=True)
        losses_t = [step(b)["loss"] for b in batches[:3]]          # 2 eager + capture/replay
        assert torch.equal(fwd(xv), yv)                            # capture + first replay of the forward
        torch.cuda.synchronize()
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for b in batches[3:]:
            with torch.cuda.stream(sa):
                losses_t.append(step(b)["loss"])
            with torch.cuda.stream(sb):
                outs.append(fwd(xv))
                outs.append(fwd(xv))
        torch.cuda.synchronize()
    assert torch.equal(torch.stack(losses_s), torch.stack(losses_t))
    for o in outs:
        assert torch.equal(o, yv)
    for (k, a), (_, b) in zip(m_s.state_dict().items(), m_t.state_dict().items()):
        assert torch.equal(a, b), k


def test_volume_feeder_with_a_consumer_that_never_synchronises():
    """the host may run subjects ahead of the GPU (no .item() in the loop).  A slot's pinned staging buffer
    must not be overwritten before the earlier H2D copy out of it has completed, and its device tensors not before the
    consumer's kernels have read them.  The consumer here only enqueues work; every volume is checked at the very end."""
    from segmentation_pipeline_amd.sampling import VolumeFeeder
    n, shape = 8, (4, 96, 96, 96)                    # 14 MB per tensor: uploads take long enough to overlap
    subs = ({"X": torch.full(shape, float(i)), "y": torch.full((1,) + shape[1:], float(-i))} for i in range(n))
    sums = torch.zeros((n, 4), device="cuda", dtype=torch.float64)
    slow = torch.randn((2048, 2048), device="cuda")
    feeder = VolumeFeeder(subs, "cuda")
    for i, vols in enumerate(feeder):
        for _ in range(20):                          # keep the GPU busy so the host gets ahead
            slow = torch.tanh(slow @ slow * 1e-3)
        sums[i, 0], sums[i, 1] = vols["X"].double().min(), vols["X"].double().max()
        sums[i, 2], sums[i, 3] = vols["y"].double().min(), vols["y"].double().max()
    torch.cuda.synchronize()
    want = torch.tensor([[i, i, -i, -i] for i in range(n)], dtype=torch.float64)
    assert torch.equal(sums.cpu(), want), sums.cpu()


def test_wsconv3d_forward_and_gradients_golden(golden):
    """WSConv3d (components.py:76-88): standardisation kernel + conv, forward and both gradients; the bias
    parameter is unused in the reference (grad None)."""
    g = golden("ensembles_ws.npz")
    ws = WSConv3d(4, 6, 3, padding=1)
    ws.load_state_dict(g.state_dict("ws.sd."))
    ws = ws.cuda()
    x = g.t("ws.x").cuda().requires_grad_()
    y = ws(x)
    assert maxerr(y, g["ws.y"]) <= 1e-4
    (y * y).sum().backward()
    grad_close(x.grad, g["ws.dx"], "ws.dx")
    grad_close(ws.weight.grad, g["ws.dw"], "ws.dw", rtol=2e-3)
    assert ws.bias.grad is None


def test_batch_stride_concat_path_n2_grads_flow(golden):
    """N=2 makes every concat slot a strided (non-contiguous) channel slice."""
    g = golden("unet_gn_convt.npz")
    model = BUILDERS["unet_gn_convt.npz"][0]()
    model.load_state_dict(g.state_dict("m.sd."))
    model = model.cuda().train()
    x = g.t("x").cuda()
    both = model(x)
    one = model(x[1:2].contiguous())
    assert maxerr(both[1:2], one.cpu()) <= 1e-6  # GroupNorm: samples are independent


@pytest.mark.parametrize("dropout_p", [0.5])
def test_dropout_train_runs_and_eval_is_identity(dropout_p):
    torch.manual_seed(0)
    m = ModularUNet(2, 2, [8, 16], 2, block_params={'dropout_p': dropout_p, **GN8}, **CONVT).cuda()
    x = torch.randn(1, 2, 8, 8, 8, device="cuda")
    m.eval()
    with torch.no_grad():
        a, b = m(x), m(x)
    assert torch.equal(a, b)
    m.train()
    p = m(x)
    assert torch.isfinite(p).all() and abs(p.sum(dim=1).mean().item() - 1.0) < 1e-5
    p.mean().backward()


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("arch", ["nested", "modular"])
def test_dropout_in_the_c8_training_flow(mode, arch):
    """Dropout3d (dmri_hippo: NestedResUNet(dropout_p=0.2), research/dmri_hippo/configs/main_config.py:123-127) in a
    16-bit mode: the channel mask is applied to the c8 activation on its way into the concat slot, forward and backward.
    With the same seed the c8 flow and the round-2 twin flow draw the same masks: probabilities agree to the mode's
    tolerance, gradients point the same way; dropped channels really are zero in effect (the result differs from p = 0)."""
    import segmentation_pipeline_amd as sp
    from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
    torch.manual_seed(3)
    if arch == "nested":
        m = NestedResUNet(3, 2, 8, dropout_p=0.3).cuda().train()
        x = torch.randn(2, 3, 16, 16, 16, device="cuda")
    else:
        m = ModularUNet(3, 2, [8, 16, 24], 3, block_params={'dropout_p': 0.3, **GN8}, **CONVT).cuda().train()
        x = torch.randn(2, 3, 16, 16, 16, device="cuda")
    y = torch.nn.functional.one_hot(torch.randint(0, 2, (2, 16, 16, 16), device="cuda"), 2).permute(0, 4, 1, 2, 3).float().contiguous()

    def run(c8only, seed=11):
        ops.H16_TRAIN_C8ONLY = c8only
        m.zero_grad()
        torch.manual_seed(seed)
        try:
            with sp.precision(mode):
                p = m(x)
                HybridLogisticDiceLoss()(p, y)["loss"].backward()
        finally:
            ops.H16_TRAIN_C8ONLY = True
        return p.detach(), torch.cat([v.grad.flatten() for v in m.parameters()]).double()
    p_c8, g_c8 = run(True)
    p_tw, g_tw = run(False)
    assert torch.isfinite(p_c8).all() and torch.isfinite(g_c8).all()
    assert (p_c8 - p_tw).abs().max().item() <= (4e-2 if mode == "bf16" else 1e-2)
    cos = float(g_c8 @ g_tw / (g_c8.norm() * g_tw.norm()))
    assert cos >= (0.98 if mode == "bf16" else 0.998), cos
    p_other, _ = run(True, seed=12)
    assert (p_c8 - p_other).abs().max().item() > 1e-3, "a different mask made no difference"


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_nested_res_unet_with_filters_not_a_multiple_of_8_keeps_the_twin_flow(mode):
    """NestedResUNet(filters=4): concat slots would not start on c8 block boundaries, so the model stays on fp32 tensors
    with c8 conv operands (the round-2 twin flow) in the 16-bit modes -- same results as fp32 within the mode's tolerance,
    gradients in the same direction."""
    import segmentation_pipeline_amd as sp
    from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
    torch.manual_seed(5)
    m = NestedResUNet(3, 2, 4).cuda().train()
    x = torch.randn(2, 3, 16, 16, 16, device="cuda")
    y = torch.nn.functional.one_hot(torch.randint(0, 2, (2, 16, 16, 16), device="cuda"), 2).permute(0, 4, 1, 2, 3).float().contiguous()

    def run(prec):
        m.zero_grad()
        with sp.precision(prec):
            p = m(x)
            HybridLogisticDiceLoss()(p, y)["loss"].backward()
        return p.detach(), torch.cat([v.grad.flatten() for v in m.parameters()]).double()
    p32, g32 = run("fp32")
    p16, g16 = run(mode)
    assert torch.isfinite(g16).all()
    assert (p16 - p32).abs().max().item() <= (4e-2 if mode == "bf16" else 1e-2)
    assert float(g16 @ g32 / (g16.norm() * g32.norm())) >= (0.97 if mode == "bf16" else 0.995)


def test_full_size_cfg2_properties():
    """BASELINE cfg2 at full size (1x4x128^3, 18.08 M params): size-independent properties."""
    torch.manual_seed(0)
    model = ModularUNet(4, 3, [32, 64, 128, 256, 320], 5, block_params=dict(GN8), **CONVT).cuda()
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn((1, 4, 128, 128, 128), generator=gen).cuda()
    lab = torch.randint(0, 3, (1, 128, 128, 128), generator=gen)
    y = torch.nn.functional.one_hot(lab, 3).permute(0, 4, 1, 2, 3).float().contiguous().cuda()
    model.eval()
    with torch.no_grad():
        p1 = model(x)
        p2 = model(x)
    assert p1.shape == (1, 3, 128, 128, 128)
    assert torch.equal(p1, p2), "forward must be bit-reproducible"
    assert torch.isfinite(p1).all() and p1.min() >= 0 and p1.max() <= 1
    assert (p1.sum(dim=1) - 1).abs().max().item() <= 1e-5
    # linearity probe of the first conv through the whole-network plumbing is not available;
    # instead: translation of the batch axis (N=1 twice == N=2) is covered at small size.
    am, counts = ops.argmax_confusion(p1, lab.to(torch.int32).cuda())
    assert counts.sum(dim=2).eq(128 ** 3).all()                      # TP+FP+FN+TN = S per class
    assert counts[0, :, 0].sum() + counts[0, :, 1].sum() == 128 ** 3  # every voxel predicted once
    model.train()
    ld = HybridLogisticDiceLoss()(model(x), y)
    assert all(torch.isfinite(v) for v in ld.values())
    assert 0.0 <= ld["dice_loss"].item() <= 1.0
    ld["loss"].backward()
    for k, v in model.named_parameters():
        assert v.grad is not None and torch.isfinite(v.grad).all(), k


def test_patch_predict_end_to_end_on_gpu(golden):
    """PatchPredict (prediction.py:124-152 counterpart): HIP tiler + model + HIP aggregator equals a
    manual tile loop with the oracle's restatement of GridSampler / GridAggregator('average')."""
    from oracle import torch_ref as R
    from segmentation_pipeline_amd.prediction import PatchPredict
    g = golden("components.npz")
    model = ModularUNet(4, 3, [8, 16], 2, block_params=dict(GN8), **CONVT)
    model.load_state_dict(g.state_dict("flips.sd."))
    model = model.cuda().eval()
    vol = torch.randn((4, 20, 16, 24), generator=torch.Generator().manual_seed(5))
    pp = PatchPredict(patch_batch_size=3, patch_size=8, patch_overlap=2)
    out = pp.predict(model, torch.device("cuda"), {"X": vol[None]})["y_pred"][0]
    locs = R.grid_locations(vol.shape[1:], (8, 8, 8), (2, 2, 2))
    with torch.no_grad():
        patches = torch.cat([model(vol[None, :, i:i + 8, j:j + 8, k:k + 8].cuda().contiguous()).cpu() for i, j, k in locs])
    ref = R.aggregate_average(patches, locs, vol.shape[1:])
    assert out.shape == (3, 20, 16, 24)
    assert maxerr(out, ref) <= 1e-5
    assert (out.sum(dim=0).cpu() - 1).abs().max() <= 1e-5  # averages of probabilities still sum to 1


def test_patch_predict_padding_mode_on_gpu(golden):
    """padding_mode='edge' / 'reflect' / a constant (prediction.py:114,132): equal to the oracle's restatement of
    GridSampler padding + GridAggregator crop (numpy.pad); parity with torchio itself stays unpinned (absent)."""
    from oracle import torch_ref as R
    from segmentation_pipeline_amd.prediction import PatchPredict
    g = golden("components.npz")
    model = ModularUNet(4, 3, [8, 16], 2, block_params=dict(GN8), **CONVT)
    model.load_state_dict(g.state_dict("flips.sd."))
    model = model.cuda().eval()
    vol = torch.randn((4, 20, 16, 24), generator=torch.Generator().manual_seed(5))
    run = lambda p: model(p.cuda()).cpu()
    for mode in ("edge", "reflect", 0.25):
        pp = PatchPredict(patch_batch_size=3, patch_size=8, patch_overlap=4, padding_mode=mode)
        out = pp.predict(model, torch.device("cuda"), {"X": vol[None]})["y_pred"][0]
        ref = R.sliding_window_average(vol, run, (8, 8, 8), (4, 4, 4), mode)
        assert out.shape == (3, 20, 16, 24)
        assert maxerr(out, ref) <= 1e-5, mode


def test_standard_predict_sagittal_split_on_gpu(golden):
    from segmentation_pipeline_amd.prediction import StandardPredict
    g = golden("components.npz")
    model = ModularUNet(4, 3, [8, 16], 2, block_params=dict(GN8), **CONVT)
    model.load_state_dict(g.state_dict("flips.sd."))
    model = model.cuda().eval()
    x = torch.randn((2, 4, 16, 8, 8), generator=torch.Generator().manual_seed(6)).cuda()
    with torch.no_grad():
        y = StandardPredict(sagittal_split=True).predict(model, torch.device("cuda"), {"X": x})["y_pred"]
        a, b = model(x[:, :, :8].contiguous()), model(x[:, :, 8:].flip(2).contiguous()).flip(2)
    assert y.shape == (2, 3, 16, 8, 8)
    assert maxerr(y, torch.cat([a, b], dim=2).cpu()) <= 1e-6


@pytest.mark.parametrize("shape,cin,cout,filters", [
    ((1, 3, 8, 32, 16), 3, 7, [8, 16, 24]),      # cfg5 family: anisotropic patch, 7 classes, odd Cin
    ((2, 2, 4, 12, 20), 2, 2, [8, 16]),          # sizes that divide no tile dimension, N = 2
])
def test_anisotropic_and_ragged_volumes_match_cpu_oracle(shape, cin, cout, filters):
    """Forward, loss and every gradient against the torch-CPU restatement of the reference
    (oracle/torch_ref.py, itself pinned to the reference goldens) on non-cubic volumes."""
    from oracle import torch_ref as R
    depth = len(filters)
    torch.manual_seed(3)
    model = ModularUNet(cin, cout, filters, depth, block_params=dict(GN8), **CONVT)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    x = torch.randn(shape, generator=g)
    lab = torch.randint(0, cout, (shape[0],) + tuple(shape[2:]), generator=g)
    y = torch.nn.functional.one_hot(lab, cout).permute(0, 4, 1, 2, 3).float().contiguous()
    spec = R.UNetSpec(cin, cout, filters, depth, norm="group", groups=8, up="convT")
    p_ref = R.unet_forward(sd, spec, x, training=True)
    ld_ref = R.hybrid_logistic_dice_loss(p_ref, y)
    ld_ref["loss"].backward()

    model = model.cuda().train()
    p = model(x.cuda())
    assert maxerr(p, p_ref.detach()) <= PROB_TOL
    ld = HybridLogisticDiceLoss()(p, y.cuda())
    assert abs(ld["loss"].item() - ld_ref["loss"].item()) <= 1e-4
    ld["loss"].backward()
    for k, v in model.named_parameters():
        grad_close(v.grad, sd[k].grad, k)


def test_device_side_samplers_on_gpu():
    from segmentation_pipeline_amd.sampling import UniformSampler, WeightedSampler
    vol = torch.randn((3, 20, 18, 22), generator=torch.Generator().manual_seed(2)).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    patches, loc = UniformSampler(8)(vol, 16, generator=g)
    for p, (i, j, k) in zip(patches, loc.tolist()):
        assert torch.equal(p, vol[:, i:i + 8, j:j + 8, k:k + 8])
    pm = torch.zeros((1, 20, 18, 22), device="cuda")
    pm[0, 10, 9, 11] = 1.0                                   # all mass on one centre
    (patches, pmaps), loc = WeightedSampler(8)(vol, pm, 5, generator=g, extra=[pm])
    assert (loc.cpu() == torch.tensor([6, 5, 7], dtype=torch.int32)).all()
    assert (pmaps[:, 0, 4, 4, 4] == 1.0).all()


def test_weighted_sampler_kernel_matches_the_cdf_restatement_and_its_distribution():
    """m355_sampler_build / m355_sampler_draw (N2): for given uniform numbers the drawn corners equal the float64
    cumulative-sum restatement of tio.WeightedSampler (oracle.torch_ref.weighted_sample_locations; torchio itself is
    absent: parity unpinned) on a ragged volume with negative / zero / huge entries; on a three-level map (background 1,
    tissue 10, lesion 100 -- transforms/image_from_labels.py) the empirical centre distribution passes a chi-square test;
    the table is built once per map version."""
    from oracle import torch_ref as R
    from segmentation_pipeline_amd.sampling import WeightedSampler
    g = torch.Generator().manual_seed(5)
    shape, patch = (37, 41, 45), (8, 6, 10)
    pm = torch.rand(shape, generator=g)
    pm[pm < 0.3] = 0.0
    pm[5:9, 7:11, 3:30] = -2.0                      # negative: counts as 0
    pm[20, 20, 20] = 5000.0
    pm[0, 0, 0] = 1e9                               # a centre whose patch does not fit: never drawn
    u = torch.rand(4096, dtype=torch.float64, generator=g)
    u[:4] = torch.tensor([0.0, 1.0 - 2.0 ** -53, 0.5, 1e-300], dtype=torch.float64)
    ref, weights = R.weighted_sample_locations(pm.numpy(), patch, u.numpy())
    table = ops.sampler_build(pm.cuda(), patch)
    loc = ops.sampler_draw(pm.cuda(), table, patch, u.cuda())
    assert torch.equal(loc.cpu(), torch.from_numpy(ref))
    assert float(table[-1]) == pytest.approx(float(weights.sum()), rel=1e-12) and float(table[0]) == 0.0
    # distribution: three levels, patch 4^3 on 24 x 20 x 28
    lv = torch.ones((24, 20, 28))
    lv[6:14, 4:12, 8:20] = 10.0
    lv[9:11, 6:9, 10:14] = 100.0
    ws = WeightedSampler(4)
    n = 200000
    dg = torch.Generator(device="cuda").manual_seed(1)
    lvd = lv.cuda()
    locs = ws.sample_locations(lvd, n, generator=dg)
    assert len(ws._tables) == 1
    ws.sample_locations(lvd, 8, generator=dg)
    assert len(ws._tables) == 1                     # cached: same storage, same version
    centre = (locs.cpu().long() + 2)
    assert (locs >= 0).all() and (locs.cpu() <= torch.tensor([20, 16, 24])).all()
    level = lv[centre[:, 0], centre[:, 1], centre[:, 2]]
    valid = torch.zeros_like(lv)
    valid[2:23, 2:19, 2:27] = lv[2:23, 2:19, 2:27]
    obs = torch.tensor([(level == v).sum().item() for v in (1.0, 10.0, 100.0)], dtype=torch.float64)
    exp = torch.tensor([valid[valid == v].sum().item() for v in (1.0, 10.0, 100.0)], dtype=torch.float64)
    exp = exp / exp.sum() * n
    chi2 = float(((obs - exp) ** 2 / exp).sum())
    assert chi2 < 13.8, (chi2, obs.tolist(), exp.tolist())      # chi-square, 2 degrees of freedom, p = 0.001
    lvd.mul_(2.0)                                   # an in-place change of the map voids the cached table
    ws.sample_locations(lvd, 8, generator=dg)
    assert len(ws._tables) == 2
    with pytest.raises(RuntimeError):
        WeightedSampler(4).sample_locations(torch.zeros((8, 8, 8), device="cuda"), 2)


def test_volume_feeder_double_buffered_uploads_on_gpu():
    """N2 feeding half: pinned, double-buffered H2D on a side stream; every subject arrives intact and in order while
    the previous one is being sampled (device-side WeightedSampler on the resident volume)."""
    from segmentation_pipeline_amd.sampling import UniformSampler, VolumeFeeder
    g = torch.Generator().manual_seed(3)
    subs = [{"X": torch.randn((2, 24, 20, 28), generator=g), "name": i} for i in range(4)]
    sampler = UniformSampler(8)
    dg = torch.Generator(device="cuda").manual_seed(0)
    for i, vols in enumerate(VolumeFeeder(subs, "cuda")):
        assert vols["name"] == i and vols["X"].is_cuda
        patches, loc = sampler(vols["X"], 6, generator=dg)
        ref = subs[i]["X"]
        for p, (a, b, c) in zip(patches.cpu(), loc.tolist()):
            assert torch.equal(p, ref[:, a:a + 8, b:b + 8, c:c + 8])


@pytest.mark.parametrize("norm", ["group", "batch"])
def test_residual_blocks_with_batch_gt1_match_cpu_oracle(norm):
    """Residual Block3d (components.py:41-46,67-68) with N = 2: the fused residual add is dense while
    the block output is a strided concat slot (regression: they need separate batch strides)."""
    from oracle import torch_ref as R
    torch.manual_seed(5)
    bp = {'residual': True}
    if norm == "group":
        bp.update(GN8)
    model = ModularUNet(3, 2, [8, 16], 2, block_params=bp, **CONVT)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k)
          for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(12)
    x = torch.randn((2, 3, 8, 8, 16), generator=g)
    lab = torch.randint(0, 2, (2, 8, 8, 16), generator=g)
    y = torch.nn.functional.one_hot(lab, 2).permute(0, 4, 1, 2, 3).float().contiguous()
    spec = R.UNetSpec(3, 2, [8, 16], 2, norm=norm, groups=8, up="convT", residual=True)
    p_ref = R.unet_forward(sd, spec, x, training=True)
    ld_ref = R.hybrid_logistic_dice_loss(p_ref, y)
    ld_ref["loss"].backward()
    model = model.cuda().train()
    p = model(x.cuda())
    assert maxerr(p, p_ref.detach()) <= PROB_TOL
    ld = HybridLogisticDiceLoss()(p, y.cuda())
    ld["loss"].backward()
    for k, v in model.named_parameters():
        grad_close(v.grad, sd[k].grad, k)


def test_derived_filters_cached_outside_autograd():
    """Blur / WS convolutions derive their filter from the parameter (box blur + rearrangement, standardisation): under
    no_grad the derived tensor is computed once per parameter version (same object on the next forward, so its packed
    forms are reused too), re-derived after an in-place update, and never cached while autograd needs its graph."""
    from segmentation_pipeline_amd.models import components as K
    torch.manual_seed(3)
    m = BlurConv3d(8, 8, 3, stride=2, padding=1).cuda()
    t = BlurConvTranspose3d(8, 8, 3, stride=2, padding=1, output_padding=0).cuda()
    ws = WSConv3d(8, 8, 3, padding=1).cuda()
    x = torch.randn(1, 8, 8, 8, 16, device="cuda")
    with torch.no_grad():
        y1, u1, s1 = m(x), t(x), ws(x)
        w1 = K._DERIVED[m][1]
        y2, u2, s2 = m(x), t(x), ws(x)
        assert K._DERIVED[m][1] is w1 and torch.equal(y1, y2) and torch.equal(u1, u2) and torch.equal(s1, s2)
        m.weight.mul_(0.5)
        y3 = m(x)
        assert K._DERIVED[m][1] is not w1 and maxerr(y3, 0.5 * y1.cpu()) < 1e-5
    y4 = m(x)                       # autograd on: derived inside the graph, gradient reaches the parameter
    y4.sum().backward()
    assert m.weight.grad is not None and torch.equal(y4.detach(), y3)
    assert "kernel" in m.state_dict() and not any("DERIVED" in k or "wexp" in k for k in m.__dict__)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_graphed_forward_and_graphed_sliding_window_are_bit_identical(golden, precision):
    """prediction.GraphedForward replays the no-grad forward from a hipGraph: the same bits as the eager forward, one
    capture per input shape, a new capture after the parameters change; PatchPredict(graph=True) == graph=False."""
    import segmentation_pipeline_amd as sp
    from segmentation_pipeline_amd.prediction import GraphedForward, PatchPredict
    g = golden("unet_gn_convt.npz")
    model = BUILDERS["unet_gn_convt.npz"][0]()
    model.load_state_dict(g.state_dict("m.sd."))
    model = model.cuda().eval()
    with sp.precision(precision):
        gf = GraphedForward(model)
        xs = [torch.randn(2, 4, 16, 16, 16, device="cuda", generator=torch.Generator("cuda").manual_seed(s)) for s in (1, 2, 3)]
        with torch.no_grad():
            eager = [model(x) for x in xs]
        for x, e in zip(xs, eager):
            assert torch.equal(gf(x), e)
        assert len(gf._graphs) == 1
        y_other = gf(torch.randn(1, 4, 8, 16, 16, device="cuda"))          # another shape: its own capture
        assert y_other.shape == (1, 3, 8, 16, 16) and len(gf._graphs) == 2
        with torch.no_grad():
            model.out_conv.bias.add_(0.5)                                   # parameters changed: captured again
            e2 = model(xs[0])
        assert torch.equal(gf(xs[0]), e2) and not torch.equal(e2, eager[0])
        vol = torch.randn(4, 40, 40, 40, generator=torch.Generator().manual_seed(5))
        outs = []
        for graph in (False, True):
            pp = PatchPredict(patch_batch_size=2, patch_size=16, patch_overlap=4, graph=graph)
            outs.append(pp.predict(model, torch.device("cuda"), {"X": vol[None]})["y_pred"][0])
        assert torch.equal(outs[0], outs[1])
        assert (outs[1].sum(dim=0) - 1).abs().max().item() <= 1e-5


# ----------------------------------------------------------------- fp16 loss scale: one cell per backward pass (round 4)
def _small_unet(cout=3, seed=0, **kw):
    torch.manual_seed(seed)
    return ModularUNet(4, cout, [8, 16], 2, block_params=dict(GN8), **CONVT, **kw).cuda().train()


def _grads(model, x, loss_fn, mode):
    import segmentation_pipeline_amd as sp
    model.zero_grad(set_to_none=True)
    with sp.precision(mode):
        loss = loss_fn(model(x))
    return loss


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-300))


def _all_cos(model, ref):
    got = torch.cat([p.grad.flatten() for _, p in model.named_parameters()])
    return _cos(got, torch.cat([ref[k].flatten() for k, _ in model.named_parameters()]))


def test_fp16_scale_is_per_backward_pass_two_models_of_different_size():
    """ADVICE r3 / VERDICT r3 5c: the fp16 loss scale was one process-wide number set by the LAST softmax out conv that ran
    forward.  Two models with different output sizes forwarded before either backward now each carry their own GradScale
    cell: their gradients equal, bit for bit, the gradients of the same passes run one after the other."""
    import segmentation_pipeline_amd as sp
    g = torch.Generator().manual_seed(3)
    xa, xb = torch.randn((1, 4, 32, 32, 32), generator=g).cuda(), torch.randn((2, 4, 8, 8, 16), generator=g).cuda()
    crit = HybridLogisticDiceLoss()

    def target(x, c):
        lab = torch.randint(0, c, (x.shape[0],) + tuple(x.shape[2:]), generator=g)
        return torch.nn.functional.one_hot(lab, c).permute(0, 4, 1, 2, 3).float().contiguous().cuda()
    ya, yb = target(xa, 3), target(xb, 3)
    ma, mb = _small_unet(seed=1), _small_unet(seed=2)
    ops.fp16_overflow()     # (clear what earlier tests may have left)
    with sp.precision("fp16"):
        # one after the other
        ma.zero_grad(set_to_none=True); mb.zero_grad(set_to_none=True)
        crit(ma(xa), ya)["loss"].backward()
        sa = ops.grad_scale(_lib_f16())
        crit(mb(xb), yb)["loss"].backward()
        sb = ops.grad_scale(_lib_f16())
        ref_a = {k: p.grad.clone() for k, p in ma.named_parameters()}
        ref_b = {k: p.grad.clone() for k, p in mb.named_parameters()}
        assert sa != sb and sa > 1 and sb > 1          # 32768 voxels vs 2 x 1024: different scales
        # interleaved: both forwards first, then the backwards in the opposite order
        ma.zero_grad(set_to_none=True); mb.zero_grad(set_to_none=True)
        la, lb = crit(ma(xa), ya)["loss"], crit(mb(xb), yb)["loss"]
        la.backward()
        lb.backward()
    for k, p in ma.named_parameters():
        assert torch.equal(p.grad, ref_a[k]), k
    for k, p in mb.named_parameters():
        assert torch.equal(p.grad, ref_b[k]), k
    assert ops.fp16_overflow() == 0


def _lib_f16():
    from segmentation_pipeline_amd import _lib
    return _lib.COMPUTE_F16


@pytest.mark.parametrize("head", ["softmax_sum_loss", "stochastic_matrix"])
def test_fp16_scale_is_calibrated_on_the_entering_gradient(head):
    """A sum-reduced criterion (gradient ~1 per voxel instead of ~1 / (N * voxels)) and the cascade's StochasticMatrix
    hypothesis (no softmax out conv: round 3 left the 2^16 default) both train in fp16: the scale comes from the gradient
    that enters the c8 flow, the gradients follow the fp32 run of the same model (cosine over all parameters >= 0.995,
    every tensor's norm within 10 %) and nothing saturates."""
    import segmentation_pipeline_amd as sp
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 4, 16, 16, 32), generator=g).cuda()
    if head == "softmax_sum_loss":
        model = _small_unet()
        w = torch.randn((1, 3, 16, 16, 32), generator=g).cuda()
        loss_fn = lambda p: (p * w).sum()                       # noqa: E731 -- sum reduction
    else:
        model = _small_unet(cout=4, hypothesis_class=StochasticMatrix, hypothesis_params={"channels": 2, "diag_bias": 5})
        w = torch.randn((1, 4, 16, 16, 32), generator=g).cuda()
        loss_fn = lambda p: (p * w).sum() / w.numel()           # noqa: E731
    _grads(model, x, loss_fn, "fp32").backward()
    ref = {k: p.grad.clone() for k, p in model.named_parameters()}
    ops.fp16_overflow()
    _grads(model, x, loss_fn, "fp16").backward()
    assert ops.fp16_overflow() == 0
    assert _all_cos(model, ref) >= 0.995
    for k, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        r = float(p.grad.double().norm() / (ref[k].double().norm() + 1e-300))
        assert 0.9 <= r <= 1.1, (k, r)
    # (the sum-reduced gradient is ~1e5 x the mean-reduced one: the size-derived scale of round 3 would have clamped it)
    scale = ops.grad_scale(_lib_f16())
    assert scale <= 2.0 ** 10 if head == "softmax_sum_loss" else scale >= 2.0 ** 8, scale


def test_fp16_overflow_is_detected_and_the_step_skipped(monkeypatch):
    """A loss scale that is far too large (forced) clamps the scaled gradients: the kernels raise the overflow word,
    trainer.train_step skips the optimizer step (weights untouched, `skipped_step` reported), and with the scale back on
    "auto" the next step recalibrates and trains."""
    import segmentation_pipeline_amd as sp
    from segmentation_pipeline_amd.prediction import StandardPredict
    from segmentation_pipeline_amd.trainer import train_step
    model = _small_unet()
    g = torch.Generator().manual_seed(9)
    x = torch.randn((1, 4, 16, 16, 16), generator=g).cuda()
    lab = torch.randint(0, 3, (1, 16, 16, 16), generator=g)
    y = torch.nn.functional.one_hot(lab, 3).permute(0, 4, 1, 2, 3).float().contiguous().cuda()
    crit, pred = HybridLogisticDiceLoss(), StandardPredict(image_names=["X", "y"])
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    with sp.precision("fp16"):
        ops.fp16_overflow()                                     # clear
        monkeypatch.setattr(ops, "FP16_GRAD_SCALE", 2.0 ** 40)
        ld, _ = train_step(model, crit, opt, pred, {"X": x, "y": y}, torch.device("cuda"))
        assert ld.get("skipped_step") is True and ops._fp16_target < 16.0
        for k, v in model.state_dict().items():
            assert torch.equal(v, before[k]), k
        monkeypatch.setattr(ops, "FP16_GRAD_SCALE", "auto")
        ld, _ = train_step(model, crit, opt, pred, {"X": x, "y": y}, torch.device("cuda"))
        assert "skipped_step" not in ld
        assert any(not torch.equal(v, before[k]) for k, v in model.state_dict().items())
        assert all(torch.isfinite(v).all() for v in model.state_dict().values())
    # bf16 never touches the word
    with sp.precision("bf16"):
        ld, _ = train_step(model, crit, opt, pred, {"X": x, "y": y}, torch.device("cuda"))
    assert not ops.fp16_overflow()
    # a FUSED optimizer takes the word as its `found_inf` tensor: the skip happens on the device, no host read in the step
    for make in (lambda ps: torch.optim.SGD(ps, lr=1e-2, momentum=0.9, fused=True), lambda ps: torch.optim.Adam(ps, lr=1e-3, fused=True)):
        model = _small_unet()
        optf = make(model.parameters())
        with sp.precision("fp16"):
            ld, _ = train_step(model, crit, optf, pred, {"X": x, "y": y}, torch.device("cuda"))   # a clean step: state exists
            assert float(ld["found_inf"]) == 0.0
            before = {k: v.clone() for k, v in model.state_dict().items()}
            state_before = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for st in optf.state.values()]
            monkeypatch.setattr(ops, "FP16_GRAD_SCALE", 2.0 ** 40)
            ld, _ = train_step(model, crit, optf, pred, {"X": x, "y": y}, torch.device("cuda"))
            monkeypatch.setattr(ops, "FP16_GRAD_SCALE", "auto")
            assert float(ld["found_inf"]) == 1.0
            for k, v in model.state_dict().items():
                assert torch.equal(v, before[k]), k
            for st, st0 in zip(optf.state.values(), state_before):
                for k, v in st.items():
                    if torch.is_tensor(v) and k != "step":
                        assert torch.equal(v, st0[k]), k
            ld, _ = train_step(model, crit, optf, pred, {"X": x, "y": y}, torch.device("cuda"))
            torch.cuda.synchronize()
            assert float(ld["found_inf"]) == 0.0 and ops._fp16_target < 16.0     # (the host has learnt of the overflow by now)
            assert any(not torch.equal(v, before[k]) for k, v in model.state_dict().items())
        ops._fp16_target = 16.0
