import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_first():
    """torch's HIP runtime must come up before librala_hip's first context in a process that uses
    both (the other order leaves torch without devices: "No HIP GPUs are available"); bench.py and
    the multi-GPU runner have that order by construction, the tests get it here."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield


@pytest.fixture
def hip_ctx_factory():
    """Factory of librala_hip contexts, destroyed after the test (a C3 context holds ~30 GB of
    HBM); fails loudly when there is no GPU / library."""
    from rala_amd import hip

    made = []

    def make(device=0):
        c = hip.Context(device)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()
