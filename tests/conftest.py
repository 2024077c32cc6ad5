import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mos_path  # noqa: E402,F401  (puts mix-of-show_amd/ on sys.path -> `import mixofshow`)

GOLDEN = os.path.join(ROOT, 'tests', 'golden', 'reference_golden.pt')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import torch
    return torch.load(GOLDEN, weights_only=False)


@pytest.fixture()
def emulated_hip(monkeypatch):
    """CPU tests of the HOST orchestration: replace the kernel-backed primitives (mixofshow.hip.ops) with the
    oracle's torch emulation. Test infrastructure only — the product never imports oracle/."""
    from oracle import emu_ops
    import mixofshow.hip.ops as ops
    for name in emu_ops.EMULATED:
        monkeypatch.setattr(ops, name, getattr(emu_ops, name))
    monkeypatch.setenv('MOS_TEST_ALLOW_CPU', '1')
    yield


@pytest.fixture()
def gpu_branches(emulated_hip):
    """CPU tests of the HOST code on the branches it takes on the GPU box: on top of `emulated_hip`, `Tensor.is_cuda`
    answers True and the 'cuda' autocast queries report the CPU autocast state, so mixofshow.hip.functional and the models
    route through their kernel-backed autograd Functions (whose kernels are the oracle's emulation here)."""
    import torch
    real_enabled, real_dtype = torch.is_autocast_enabled, torch.get_autocast_dtype
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.is_autocast_enabled = lambda device_type=None: real_enabled('cpu')
    torch.get_autocast_dtype = lambda device_type=None: real_dtype('cpu')
    try:
        yield
    finally:
        del torch.Tensor.is_cuda                      # the C-level descriptor of the base class is visible again
        torch.is_autocast_enabled, torch.get_autocast_dtype = real_enabled, real_dtype
