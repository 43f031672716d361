"""pytest config: `-m gpu` tests need a real MI355X (run through gpurun / the driver);
everything else runs on CPU.  The oracle (oracle/) is test infrastructure and is
built on demand."""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def _have_gpu() -> bool:
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run via gpurun)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_first():
    """On a GPU box, initialise torch's HIP runtime before libvkindex makes its first HIP call: torch ships its own
    copy of the runtime, and when the system runtime (libvkindex links /opt/rocm) initialises first, torch's later
    initialisation reports "No HIP GPUs are available" (seen on a fresh box with a test that touched torch only after
    building an index).  The other order -- what bench.py does -- works."""
    if _have_gpu():
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    return O


def reference_vectors(size, dims, max_value):
    """testing/common.cc:42-53 DeterministicallyGenerateVectors (f32 arithmetic)."""
    import numpy as np
    i = np.arange(size, dtype=np.float32)[:, None]
    j = np.arange(dims, dtype=np.float32)[None, :]
    return (np.float32(max_value) * ((i + j) / np.float32(size + dims))).astype(np.float32)


def timing_bound(attempts=3):
    """For tests that assert TIME bounds (a cancelled caller back within milliseconds, a stopped pass within 200 ms) on threads
    of a container with a CPU quota: a quota just spent on compiles or on a 16-thread graph build stalls a thread for tens of
    milliseconds now and then.  The bound must fail `attempts` times in a row to fail the test."""
    import functools
    import time

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*a, **kw):
            for i in range(attempts):
                try:
                    return fn(*a, **kw)
                except AssertionError:
                    if i == attempts - 1:
                        raise
                    time.sleep(0.5)
        return wrapper
    return deco
