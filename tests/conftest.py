import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A test that hangs (a rank waiting at a collective, a kernel that never ends) must FAIL with its name, not leave the whole
    run silent until the GPU box's watchdog (420 s without output) kills it: a per-test limit below that, where pytest-timeout
    is installed (the slowest legitimate test, the two-rank config-5 rehearsal, takes about a minute)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(390))


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def randomize_bn_(module, seed):
    """The BatchNorm randomisation make_golden.py applies before its eval-mode fixtures (same generator, same draw order)."""
    import torch
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                m.running_mean.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.bias.shape, generator=g) + 0.5)


def g14_model_and_batch():
    """Fixture G14's model (built on the CPU: torch.manual_seed(0) + ctor gives the reference's weights) and batch."""
    import torch
    import unet_amd
    r = load_golden("g14_eval_full_unet_512")
    torch.manual_seed(0)
    model = unet_amd.UNet(1, 1, bilinear=True)
    randomize_bn_(model, 14)
    with torch.no_grad():
        model.outc.conv.bias.copy_(torch.from_numpy(r["outc_bias"]))
    g = torch.Generator().manual_seed(1400)
    images = torch.rand(2, 1, 512, 512, generator=g)
    masks = torch.randint(0, 3, (2, 512, 512), generator=g)
    return r, model, images, masks
