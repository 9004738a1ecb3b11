import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def ew_err(a, b):
    """element-wise parity metric of the north-star tensors: max over elements of |a - b| / (|b| + rms(b)).  ``ew_err(a, b) < 1e-4`` is
    |a - b| <= 1e-4 |b| + 1e-4 rms(b) for EVERY element -- small components (tx, ty beside Tz; joints near the origin) are held to the tensor's
    own scale instead of hiding behind its largest entry."""
    import torch
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    rms = b.pow(2).mean().sqrt().clamp_min(1e-30)
    return ((a - b).abs() / (b.abs() + rms)).max().item()


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    return torch.device('cuda:0')


@pytest.fixture(scope='session')
def assets():
    from oracle import synth
    return synth.make_assets(0)


@pytest.fixture(scope='session')
def state_dict(assets):
    from oracle import synth
    return synth.make_state_dict(0, assets)
