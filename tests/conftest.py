import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _fp32_precision_after_gpu_test(request):
    """st_set_precision is a process-wide switch of the HIP library: whatever a GPU test (or an engine it created) left it at,
    the next test starts from the fp32 parity arithmetic again."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        try:
            from signaltrain_amd import _lib
            _lib.load().st_set_precision(0)
        except Exception:
            pass
