import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `-m gpu`)")


@pytest.fixture(scope="session", autouse=True)
def _build_everything():
    """Build the product library, the CPU oracle and the host twin once per session (all cached)."""
    from lajolla_public_amd import build
    build.build_product(verbose=False)
    build.build_oracle(verbose=False)
    build.build_twin(verbose=False)
