import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU-marked tests skip (instead of failing with 'No HIP GPUs') on a box without a GPU."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="needs a ROCm GPU (torch.cuda.is_available() is False)")
    for item in items:
        if item.get_closest_marker("gpu") is not None:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def tuning_state_is_at_the_shipped_defaults():
    """The library's diagnostic switches (st_set_tuning / st_set_debug) are process-wide: every test must leave them at the shipped defaults
    (st_tuning_defaults), and the defaults themselves are frozen in tests/test_abi_and_host.py::test_tuning_defaults_are_frozen."""
    yield
    import ctypes as C
    from signaltrain_amd import _lib
    try:
        lib = _lib.load()
    except Exception:
        return
    n = lib.st_get_tuning(None, 0)
    cur, dflt = (C.c_int * n)(), (C.c_int * n)()
    lib.st_get_tuning(cur, n); lib.st_tuning_defaults(dflt, n)
    leaked = [(i, cur[i], dflt[i]) for i in range(n) if cur[i] != dflt[i]]
    lib.st_reset_tuning()
    assert not leaked, f"a test left diagnostic switches set (index, value, default): {leaked}"
