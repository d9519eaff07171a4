import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _build_native_if_missing():
    """the suites assume the in-tree libraries exist (the driver's build() makes them); build them if a fresh checkout has none"""
    import subprocess
    libs = [os.path.join(ROOT, "grappa_amd", n) for n in ("libgrappa_hip.so", "libgrappa_host.so")]
    if not all(os.path.exists(p) for p in libs):
        subprocess.run(["make", "-C", os.path.join(ROOT, "grappa_amd", "csrc"), "-j8"], check=True)


def pytest_configure(config):
    _build_native_if_missing()
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


@pytest.fixture(autouse=True)
def _library_defaults(request):
    """GPU tests: every test starts from the library's default plan settings (split-K tail launches on; a model on four head streams turns
    them off for the products that follow, which would otherwise leak into the product-level tests behind it)"""
    if request.node.get_closest_marker("gpu") is not None:
        from grappa_amd import backend
        be = backend._BACKEND
        if be is not None and hasattr(be, "pin_tail_launches"):
            be.pin_tail_launches(None)
            be._tails = None
            be.plan_override = None
            be.splitk_reduce = 0
            # ... and from the default arithmetic and storage (a test that switches them and fails before restoring must not move the rest)
            from grappa_amd import ops
            default = os.environ.get("GRAPPA_GEMM_PRECISION", backend.DEFAULT_GEMM_PRECISION)
            if be.gemm_precision_name != default:
                be.set_gemm_precision(default)
            be.set_gemm_precision_bwd(None)
            ops.set_activation_dtype("f32")
    yield
