import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def _have_gpu() -> bool:
    """True when a HIP device is visible.  A box with a GPU node (/dev/kfd) but no built library counts as "have": the
    gpu tests must then FAIL loudly (no silent skip, no CPU fallback), not disappear."""
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        import halo2_amd
        return halo2_amd.lib().h2_device_count() > 0
    except Exception:
        return True


def pytest_collection_modifyitems(config, items):
    if any(item.get_closest_marker("gpu") for item in items) and not _have_gpu():
        skip = pytest.mark.skip(reason="needs an MI355X (no HIP device visible)")
        for item in items:
            if item.get_closest_marker("gpu"):
                item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
