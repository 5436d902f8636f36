import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def tiny_net():
    """(plan, oracle torch net, numpy state dict) of the committed tiny network."""
    import numpy as np
    import torch
    from deepwmh_amd.plan import tiny_plan
    from oracle.generic_unet import unet_from_plan
    plan = tiny_plan(patch=32, base=4, num_pool=3, max_feat=32)
    sd = dict(np.load(os.path.join(GOLDEN, 'tiny_net_weights.npz')))
    net = unet_from_plan(plan).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return plan, net, sd


@pytest.fixture(scope='session')
def tiny_io():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, 'tiny_net_io.npz')))
