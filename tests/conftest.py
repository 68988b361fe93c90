import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # no test of this suite runs longer than a minute or two (a fresh box may page the image in for another minute or two): a hang (worker pool, device) must fail, not eat the GPU box's time
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 900


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library; GPU tests fail loudly (not skip) if it is missing."""
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    from denet_amd import lib
    return lib.load()
