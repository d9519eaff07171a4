import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def ref_backend():
    """Test-only backend (oracle/ops_ref.py: torch restatement of the C ABI) so that the product's host logic
    runs on a machine without a GPU.  The product itself never falls back to it."""
    from grappa_amd import backend
    from oracle.ops_ref import RefBackend
    old = backend._BACKEND
    backend.set_backend(RefBackend())
    yield backend.get_backend()
    backend.set_backend(old)
