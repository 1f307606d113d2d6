import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before anything starts the HIP runtime: see syncvsr_amd/__init__.py

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _tuning_overrides():
    """SVSR_TEST_TUNE="key=value,key=value": run the suite with non-default (result-preserving) tuning knobs, e.g. to put an
    experimental kernel variant under the same parity tests as the default one."""
    spec = os.environ.get("SVSR_TEST_TUNE", "")
    if spec:
        from syncvsr_amd import ops

        for kv in spec.split(","):
            k, v = kv.split("=")
            ops.tune(k, int(v))
    yield
