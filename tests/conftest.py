import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A ceiling per test when pytest-timeout is installed (it is in this image): the whole GPU suite takes two
    minutes, so a test that is still running after ten is stuck in something outside Python's reach (a collective's
    rendezvous, a driver call) and the run should fail there instead of sitting until the caller's limit."""
    # Collection order on the GPU box (the driver runs -x): the cheap, high-coverage parity files first, the long
    # subprocess campaigns (fuzz, the reference's mains, full-size graphs) last, so that a failure in a campaign
    # cannot leave the per-op / golden / kernel tests unexecuted.  Stable within a file.
    late = {"test_gpu_reftests": 1, "test_gpu_dropin": 2, "test_gpu_fullsize": 3, "test_gpu_fuzz": 4}
    items.sort(key=lambda it: late.get(os.path.splitext(os.path.basename(str(it.fspath)))[0], 0))
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            # GPU tests: the "thread" method -- a test stuck inside a driver call never returns to the interpreter, so
            # the default (a signal handled between bytecodes) would never fire; the watchdog thread dumps the stacks and
            # ends the run instead
            if item.get_closest_marker("gpu") is not None:
                item.add_marker(pytest.mark.timeout(600, method="thread"))
            else:
                item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip():
    """The product package; on a GPU box a missing/unloadable HIP library is a failure,
    never a skip."""
    import graphblast_amd
    return graphblast_amd
