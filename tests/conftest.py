import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "whisper-finetune_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_logmel():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "logmel_golden.npz"))


@pytest.fixture(scope="session")
def golden_augment():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "augment_golden.npz"))
