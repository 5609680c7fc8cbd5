import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle (torch fp32) runs inside many tests.  torch sizes its thread pool by the HOST's core count, while a GPU box gives
    # one GPU's share of cores (16): on a many-core host the oversubscribed pool made the same suite take 400 s instead of 20 s.
    try:
        import torch
        n = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        torch.set_num_threads(max(1, n))
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
