"""Host-side logic of the drop-in surface (no GPU): constructor signatures, state_dict keys,
seeded initialisation, plug-in validation, loud failure on CPU tensors, pickling."""
import inspect
from functools import partial

import dill
import numpy as np
import pytest
import torch
from torch import nn

from segmentation_pipeline_amd import _lib
from segmentation_pipeline_amd.models import (Block3d, BlurConv3d, BlurConvTranspose3d, EnsembleFlips,
                                              EnsembleModels, EnsembleOrientations, ModularUNet,
                                              NestedResUNet, StochasticMatrix, WSConv3d)
from segmentation_pipeline_amd.models.ensemble import apply_strategy
from segmentation_pipeline_amd.models.utils import filter_kwargs
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss

GN8 = {'normalization_class': partial(nn.GroupNorm, 8)}
CONVT = dict(upsample_class=nn.ConvTranspose3d, upsample_params={'kernel_size': 2, 'stride': 2})

BUILDERS = {
    "unet_default_bn.npz": lambda: ModularUNet(4, 3, [8, 16, 32], 3),
    "unet_gn_convt.npz": lambda: ModularUNet(4, 3, [8, 16, 32], 3, block_params=dict(GN8), **CONVT),
    "unet_res_blur.npz": lambda: ModularUNet(
        2, 2, [8, 8, 16], 3, block_params={'residual': True}, downsample_class=BlurConv3d,
        downsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1}, upsample_class=BlurConvTranspose3d,
        upsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1, 'output_padding': 0}),
    "nested_res_unet.npz": lambda: NestedResUNet(3, 2, 8),
}


@pytest.mark.parametrize("name", list(BUILDERS))
def test_state_dict_keys_shapes_and_seeded_init_match_reference(golden, name):
    """Same sub-module names AND creation order as the reference: a model built here under
    torch.manual_seed(0) must equal the reference's state_dict bit for bit."""
    g = golden(name)
    ref_sd = g.state_dict("m.sd.")
    torch.manual_seed(0)
    model = BUILDERS[name]()
    sd = model.state_dict()
    assert list(sd.keys()) == list(ref_sd.keys())
    for k in sd:
        assert sd[k].shape == ref_sd[k].shape, k
        assert torch.equal(sd[k], ref_sd[k]), f"seeded init differs at {k}"
    model.load_state_dict(ref_sd)  # reference checkpoints load strictly


def test_cfg2_architecture_param_count_and_init(golden):
    g = golden("cfg2_arch_32cube.npz")
    torch.manual_seed(0)
    model = ModularUNet(4, 3, [32, 64, 128, 256, 320], 5, block_params=dict(GN8), **CONVT)
    assert sum(p.numel() for p in model.parameters()) == int(g["n_params"]) == 18080419
    # (the double sums are thread-count dependent in the last bits; the weights themselves are exact)
    np.testing.assert_allclose([p.double().sum().item() for p in model.parameters()], g["param_sums"], rtol=1e-10, atol=1e-10)


def test_constructor_signatures_mirror_reference():
    assert list(inspect.signature(ModularUNet.__init__).parameters)[1:] == [
        "in_channels", "out_channels", "filters", "depth", "block_class", "block_params", "upsample_class",
        "upsample_params", "downsample_class", "downsample_params", "out_conv_class", "out_conv_params",
        "hypothesis_class", "hypothesis_params"]
    assert list(inspect.signature(Block3d.__init__).parameters)[1:] == [
        "in_channels", "out_channels", "conv_class", "conv_params", "normalization_class", "normalization_params",
        "activation_class", "activation_params", "residual", "residual_params", "dropout_p", "num_convs"]
    assert list(inspect.signature(NestedResUNet.__init__).parameters)[1:] == [
        "input_channels", "output_channels", "filters", "dropout_p", "hypothesis_class", "hypothesis_params"]
    sig = inspect.signature(HybridLogisticDiceLoss.__init__).parameters
    assert (sig["dice_weight"].default, sig["logistic_class_weights"].default, sig["square_dice"].default) == (0.5, None, True)


def test_filters_int_and_mismatch():
    m = ModularUNet(1, 2, 8, 2)
    assert m.down_blocks[1].layers.conv0.weight.shape == (8, 8, 3, 3, 3)
    with pytest.raises(ValueError, match="does not match depth"):
        ModularUNet(1, 2, [8, 16], 3)


def test_unsupported_plugins_fail_loudly_at_construction():
    with pytest.raises(NotImplementedError, match="activation_class"):
        Block3d(4, 8, activation_class=nn.Tanh, activation_params={})
    with pytest.raises(NotImplementedError, match="normalization_class"):
        Block3d(4, 8, normalization_class=nn.LayerNorm)
    with pytest.raises(NotImplementedError, match="conv_class"):
        Block3d(4, 8, conv_class=nn.Conv2d)
    with pytest.raises(NotImplementedError, match="block_class"):
        ModularUNet(1, 2, [4, 8], 2, block_class=lambda i, o: nn.Conv3d(i, o, 3, padding=1))


def test_cpu_tensors_are_rejected_no_fallback():
    m = ModularUNet(4, 3, [8, 16], 2, block_params=dict(GN8), **CONVT)
    with pytest.raises(_lib.M355Error, match="no CPU fallback"):
        m(torch.randn(1, 4, 8, 8, 8))
    with pytest.raises(_lib.M355Error, match="no CPU fallback"):
        HybridLogisticDiceLoss()(torch.rand(1, 2, 4, 4, 4), torch.rand(1, 2, 4, 4, 4))
    with pytest.raises(RuntimeError, match="square of the number"):
        StochasticMatrix(2)(torch.randn(1, 3, 2, 2, 2))


def test_blur_buffers_match_reference(golden):
    g = golden("components.npz")
    torch.manual_seed(1)
    bc = BlurConv3d(8, 8, 3, stride=2, padding=1)
    bt = BlurConvTranspose3d(8, 8, 3, stride=2, padding=1, output_padding=0, weight_standardization=True)
    assert torch.equal(bc.kernel, g.t("blur.sd.kernel")) and float(bc.kernel.flatten()[0]) == 1 / 64
    assert torch.equal(bt.kernel, g.t("blurT.sd.kernel")) and float(bt.kernel.flatten()[0]) == 1 / 8
    assert torch.equal(bc.weight, g.t("blur.sd.weight"))  # same RNG consumption
    # effective (blurred) filter == F.conv3d(weight, kernel, padding=1, groups=C)
    w_eff, b_eff = bc.effective()
    ref = torch.nn.functional.conv3d(bc.weight, bc.kernel, padding=1, groups=8)
    assert b_eff is None and w_eff.shape == (8, 8, 4, 4, 4)
    torch.testing.assert_close(w_eff, ref, rtol=1e-6, atol=1e-7)
    # WSConv3d's standardisation is a HIP kernel (m355_weight_standardize_fwd): no torch / CPU fallback
    from segmentation_pipeline_amd._lib import M355Error
    with pytest.raises(M355Error, match="no CPU fallback"):
        WSConv3d(4, 6, 3, padding=1).effective()


def test_apply_strategy_matches_reference(golden):
    g = golden("components.npz")
    preds = list(g.t("ens.preds"))
    torch.testing.assert_close(apply_strategy(preds, "mean"), g.t("ens.mean"), rtol=0, atol=0)
    assert torch.equal(apply_strategy(preds, "majority"), g.t("ens.majority"))
    with pytest.raises(ValueError):
        EnsembleModels([nn.Identity()], strategy="median")
    assert len(EnsembleFlips(nn.Identity()).flips) == 8
    e = EnsembleOrientations(nn.Identity())
    assert len(e.permutations) * len(e.flips) == 48


def test_ensembles_with_pointwise_member():
    """flip / permute bookkeeping: with a pointwise member every ensemble is the identity."""
    class Pointwise(nn.Module):
        def forward(self, x):
            return torch.softmax(x * 2.0, dim=1)
    x = torch.randn(1, 3, 4, 5, 6)
    ref = Pointwise()(x)
    torch.testing.assert_close(EnsembleFlips(Pointwise())(x), ref)
    torch.testing.assert_close(EnsembleOrientations(Pointwise())(x), ref)
    torch.testing.assert_close(EnsembleModels([Pointwise(), Pointwise()])(x), ref)


def test_filter_kwargs():
    assert filter_kwargs(nn.ConvTranspose3d, in_channels=1, out_channels=2, channels=3) == {
        "in_channels": 1, "out_channels": 2}
    assert filter_kwargs(nn.AvgPool3d, in_channels=1, channels=3) == {}


def test_components_are_picklable_by_reference_for_torchcontext():
    """TorchContext stores the constructor itself via dill (utils/torch_context.py:210-220)."""
    for cls in (ModularUNet, Block3d, NestedResUNet, HybridLogisticDiceLoss, BlurConv3d):
        assert dill.loads(dill.dumps(cls)) is cls
        assert inspect.getsourcefile(cls)


def test_reference_format_checkpoint_roundtrip(golden, tmp_path):
    """Checkpoint interchange in the reference's on-disk format (utils/torch_context.py:113-126,
    196-213): a dict with `component_definitions` = [{name, constructor, params, state_dict}], written
    with torch.save(pickle_module=dill).  The file here is written by this test (the constructor
    pickled by reference is ours, the weights are the reference-generated golden state_dict with the
    reference's key names) and re-initialised exactly as TorchContext._init_component does."""
    g = golden("unet_res_blur.npz")
    params = dict(in_channels=2, out_channels=2, filters=[8, 8, 16], depth=3, block_params={'residual': True},
                  downsample_class=BlurConv3d, downsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1},
                  upsample_class=BlurConvTranspose3d,
                  upsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1, 'output_padding': 0})
    definitions = [
        dict(name="model", constructor=ModularUNet, params=params, state_dict=g.state_dict("m.sd.")),
        dict(name="criterion", constructor=HybridLogisticDiceLoss, params={}),
        dict(name="optimizer", constructor=torch.optim.SGD,
             params=dict(params="self.model.parameters()", lr=0.001, momentum=0.95)),
    ]
    path = tmp_path / "checkpoint.pt"
    torch.save(dict(name="ctx", component_definitions=definitions, creation_time="t", variables=None,
                    file_paths=[], metadata={}, config={}), path, pickle_module=dill)
    ckpt = torch.load(path, pickle_module=dill, weights_only=False)  # our own file, see docstring
    built = {}
    for d in ckpt["component_definitions"]:
        p = {k: (built["model"].parameters() if v == "self.model.parameters()" else v) for k, v in d["params"].items()}
        comp = d["constructor"](**p)
        if "state_dict" in d:
            comp.load_state_dict(d["state_dict"])  # strict: key names and shapes are the reference's
        built[d["name"]] = comp
    assert type(built["model"]) is ModularUNet and type(built["criterion"]) is HybridLogisticDiceLoss
    sd = built["model"].state_dict()
    ref = g.state_dict("m.sd.")
    assert list(sd) == list(ref)
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k
    assert len(built["optimizer"].param_groups[0]["params"]) == len(list(built["model"].parameters()))


def test_graphed_forward_is_transparent_off_the_gpu():
    """prediction.GraphedForward only captures on the GPU: with a CPU tensor it is the plain call (the CPU double of the
    distributed tests goes through PatchPredict(graph=True) unchanged)."""
    from segmentation_pipeline_amd.prediction import GraphedForward
    lin = torch.nn.Conv3d(2, 3, 1)
    gf = GraphedForward(lin)
    x = torch.randn(1, 2, 3, 4, 5)
    assert torch.equal(gf(x), lin(x)) and not gf._graphs


def test_precision_modes_map_to_the_abi_compute_codes():
    """"fp32" (the default) asks the library for M355_COMPUTE_F32X3 -- fp32 tensors and results, the wide convolutions on the
    bf16 matrix pipe through the exact three-way operand split; "fp32_mfma" for M355_COMPUTE_F32; the 16-bit modes for
    theirs.  Unknown names are refused."""
    import pytest
    from segmentation_pipeline_amd import _lib, ops
    assert (_lib.COMPUTE_F32, _lib.COMPUTE_BF16, _lib.COMPUTE_F16, _lib.COMPUTE_F32X3) == (0, 1, 2, 3)
    assert ops.get_precision() == "fp32" and ops.is_fp32()
    assert ops._COMPUTE["fp32"] == (_lib.COMPUTE_F32X3 if ops.FP32_SPLIT else _lib.COMPUTE_F32)
    assert ops._COMPUTE["fp32_mfma"] == _lib.COMPUTE_F32
    with ops.precision("bf16"):
        assert not ops.is_fp32() and ops.get_precision() == "bf16"
    with ops.precision("fp32_mfma"):
        assert ops.is_fp32()
    assert ops.get_precision() == "fp32"
    with pytest.raises(ValueError):
        ops.set_precision("fp32x3")
