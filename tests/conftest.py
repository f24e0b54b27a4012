import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import oracle as O
    O.load()
    return O


@pytest.fixture(scope="session")
def gpu_ctx_strict():
    from athenapk_amd import hydro
    ctx = hydro.Context(strict=True)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def gpu_ctx_fast():
    from athenapk_amd import hydro
    ctx = hydro.Context(strict=False)
    yield ctx
    ctx.close()
