import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library must exist for every test session (CPU tests only load it; GPU tests call it)."""
    import protoasnet_amd

    if not os.path.exists(protoasnet_amd.lib_path()):
        protoasnet_amd.build_extension()
    yield


def assert_close(actual, expected, atol, rtol=0.0, name=""):
    a = torch.as_tensor(actual).detach().float().cpu()
    e = torch.as_tensor(expected).detach().float().cpu()
    assert a.shape == e.shape, f"{name}: shape {tuple(a.shape)} != {tuple(e.shape)}"
    err = (a - e).abs()
    tol = atol + rtol * e.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax(err - tol))
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{bad.numel()} elements off; worst |{a.flatten()[i]:.6g} - {e.flatten()[i]:.6g}| = "
            f"{err.flatten()[i]:.3g} > {tol.flatten()[i] if tol.numel() > 1 else float(tol):.3g}"
        )
