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


AB_LIB = os.path.join(ROOT, "build", "ab", "libhalo2_mi355x_ab.so")


def ab_env(**switches):
    """Environment of a CHILD process that loads the laboratory build (`make -C halo2_amd/csrc ab`: the same sources with -DH2_AB=1, where
    the A/B switches of csrc/common.h's ab_env() are live) with the given switches set.  The shipped library reads none of them, so a
    test of a switched arm must run against that build -- through halo2_amd (H2_LIB_PATH) or the native driver (H2BENCH_LIB)."""
    if not os.path.exists(AB_LIB):
        pytest.skip(f"{AB_LIB} not built (run __graft_entry__.build())")
    return dict(os.environ, H2_LIB_PATH=AB_LIB, H2BENCH_LIB=AB_LIB, **switches)


def run_test_in_ab_child(test_file, k_expr, expect="1 passed", **switches):
    """Runs `pytest test_file -k k_expr` in a child process on the laboratory build with the switches set (they are read once per process)."""
    import subprocess
    env = ab_env(**switches)
    env["H2_AB_CHILD"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", test_file, "-k", k_expr], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and expect in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
