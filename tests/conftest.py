import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """torch bundles its own HIP runtime: it has to be initialised BEFORE libsvtvp9_hip.so pulls in the system one,
    otherwise torch reports "No HIP GPUs are available" later in the same process (GPU tests that allocate device
    buffers with torch, bench.py does the same)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # no torch / no GPU: CPU-only run
        pass
