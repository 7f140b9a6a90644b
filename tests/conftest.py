import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The engines zero the flat gradient INSIDE their optimizer sweep (hs_adamw_fused: HF Trainer's model.zero_grad() after every step).  Most
# tests here inspect gradients after a full train_step: they ask every engine of the process to keep them (engine._Engine.keep_grads).
# The fused-zeroing default itself is covered by tests that set keep_grads=False explicitly and compare the two paths bit for bit
# (tests/test_gpu_engine.py::test_fused_optimizer_sweep_equals_keep_grads_path).
os.environ.setdefault("HS_KEEP_GRADS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this container")
    from hsenet_amd import _lib
    _lib.load()                       # fail loudly if the HIP extension is missing
    assert _lib.load().hs_device_check(0) == 0, "not a gfx950 device"
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load
