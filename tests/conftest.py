import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: takes more than a few seconds on CPU')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed_every_test():
    """Every test starts from the same generator state (torch CPU + device, numpy): a test that draws inputs without
    seeding them itself — and holds a tolerance that is a statement about round-off, ReLU flips or pixel-boundary
    points — must not pass or fail by the draw."""
    import numpy as np
    import torch
    torch.manual_seed(20240917)
    np.random.seed(20240917 % (2 ** 32))
    yield
