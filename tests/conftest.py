import os as _os

# one hardware queue per stream-priority level: the configuration the codec objects are measured in; set here, in
# front of the first CUDA / HIP call of the test process, so that the plug-in finds it (it warns when the runtime is up without it)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The product library and the oracle are built in-tree before any test runs (no-op when the
    prebuilt files are current; on the GPU box the prebuilt .so files travel with the snapshot)."""
    from dcvc_amd import build as product_build
    from oracle import build_oracle
    product_build.build()
    build_oracle.build_liboracle()
    build_oracle.build_ref()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
