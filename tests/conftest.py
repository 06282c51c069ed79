import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: asserts a rate floor (GPU box); not part of the parity run - select with -m 'gpu and perf'")


def pytest_collection_modifyitems(config, items):
    """Rate floors are not parity: on a loaded box a perf wobble must not turn the parity run red.  Tests marked `perf` run only
    when the marker expression names them (-m "gpu and perf")."""
    if "perf" in (config.option.markexpr or ""):
        return
    skip = pytest.mark.skip(reason="rate floor: run with -m 'gpu and perf'")
    for it in items:
        if "perf" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); built on demand with gcc."""
    so = os.path.join(ROOT, "oracle", "librir_oracle.so")
    src = os.path.join(ROOT, "oracle", "rir_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "librir_oracle.so"])
    from oracle.pyoracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The compiled reference (oracle/_ref) - only where it was built."""
    from oracle.pyoracle import Ref

    if not Ref.available():
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    return Ref()


@pytest.fixture(scope="session")
def golden():
    d = os.path.join(ROOT, "tests", "golden")
    arrays = np.load(os.path.join(d, "signal_processing.npz"))
    with open(os.path.join(d, "signal_processing_sha256.json")) as f:
        hashes = json.load(f)
    return arrays, hashes


@pytest.fixture(scope="session")
def lib():
    """The product shared object (must exist: built by __graft_entry__.build())."""
    from librir_amd.low_level.misc import _lib

    return _lib


@pytest.fixture(scope="session")
def dev():
    """Device batch API; GPU tests only."""
    import torch

    from librir_amd import device

    assert torch.cuda.is_available(), "GPU test started without a GPU"
    assert device.device_available()
    return device
