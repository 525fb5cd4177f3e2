import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def _usable_cpus():
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


@pytest.fixture(autouse=True, scope="session")
def _oracle_threads():
    """The CPU oracle runs on torch's intra-op pool: size it to the CPUs this process may actually use (a GPU box exposes far more
    hardware threads than its share; oversubscribing them makes the Hiera-L oracle several times slower)."""
    import torch
    torch.set_num_threads(max(1, min(_usable_cpus(), 32)))
    yield
