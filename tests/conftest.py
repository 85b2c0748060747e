import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-generative_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    # PG_GUARD=1: every device tensor of the run in its own mapping between unmapped guard pages + canaries
    # (tests/guard/): out-of-bounds accesses of the kernels fault at the launch or fail the test that made them.
    import guard

    if guard.enabled():
        guard.install()


@pytest.fixture(autouse=True)
def _guard_canaries(request):
    import guard

    if not guard.enabled():
        yield
        return
    before, _ = guard.check_all()
    yield
    after, report = guard.check_all()
    if after != before:
        pytest.fail(f"guard allocator: {after - before} out-of-bounds write(s) during this test:\n{report}")


@pytest.fixture(scope="session")
def lib():
    from pytorch_generative_amd import _lib

    return _lib.load()
