import os
import sys

import pytest

# PyTorch-ROCm ships its own HIP runtime.  Load it before libdabx.so pulls in /opt/rocm's (same SONAME, the loader then
# shares one instance), as bench.py and __graft_entry__.smoke() do: tests that build their input on the device with torch
# (test_gpu_fullsize.py) must see the very runtime the library allocates with, whatever order the test files run in.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_perf: needs a real MI355X and asserts a throughput ratio (not part of the parity suite: pytest -m 'gpu or gpu_perf')")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libdabref.so (genuine reference objects)")


def pytest_collection_modifyitems(config, items):
    # gpu_perf tests are not `gpu` tests (the driver's `-m gpu -x` parity run leaves them out), so `-m "not gpu"` on a box without a GPU
    # collects them: skipped there
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="gpu_perf: needs a real MI355X")
    for it in items:
        if "gpu_perf" in it.keywords:
            it.add_marker(skip)
