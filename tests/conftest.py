import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_slow: GPU tests kept out of `-m gpu` to hold the driver's suite under 10 minutes (run with -m gpu_slow; "
                                       "tools/gpu_run.sh gpuslow); every one has a faster sibling in `-m gpu`")
    # The CPU oracle (torch fp32) is the checker of most GPU tests.  On the GPU boxes torch defaults to one thread per hardware thread
    # (256): the ViT-L oracle then runs ~100x SLOWER than at 32 threads (measured: 239 s vs 2.2 s per 512x512 view), which turned the
    # GPU suite into 15 minutes of CPU oversubscription.  Cap it at the count bench.py's cpu_baseline sweep finds fastest on those boxes
    # (8 threads 2.2 s per view, 16 -> 1.35, 32 -> 1.73).
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def built_lib():
    """The C-ABI library must exist for both CPU (symbol/argument checks) and GPU tests: build it if missing."""
    from fast3r_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def pytest_collection_modifyitems(config, items):
    """`gpu_slow` tests also carry their module's `gpu` marker: keep them out of a plain `-m gpu` run (the driver's), in unless asked for"""
    if "gpu_slow" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="gpu_slow: run with -m gpu_slow (tools/gpu_run.sh gpuslow)")
    for it in items:
        if it.get_closest_marker("gpu_slow"):
            it.add_marker(skip)
