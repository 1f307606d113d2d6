import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before anything starts the HIP runtime: see syncvsr_amd/__init__.py

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# SVSR_REDZONE=1: every device tensor of this pytest process gets poisoned 4 KiB red zones on both sides (tests/redzone/redzone_alloc.cpp,
# installed as torch's device allocator BEFORE the first allocation) and every test ends with a sweep that fails it if a kernel wrote
# into one (tests/test_gpu_redzone.py runs the kernel-level test files this way).
# SVSR_TAILFLUSH=1: out-of-bounds READS — every device tensor is its own virtual-memory reservation that ends flush (to 16 bytes) against an
# UNMAPPED page (same source file, tf_malloc): the first 16-byte access behind a tensor is a GPU memory-access fault that aborts the process.
_RZ = None
_TF = None
if os.environ.get("SVSR_TAILFLUSH") == "1":
    import ctypes
    import subprocess
    import tempfile

    import torch

    _src = os.path.join(ROOT, "tests", "redzone", "redzone_alloc.cpp")
    _so = os.path.join(tempfile.mkdtemp(prefix="svsr_tf_"), "libredzone.so")
    subprocess.run(["hipcc", "-shared", "-fPIC", "-O2", "-w", "-o", _so, _src], check=True, capture_output=True)
    torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(_so, "tf_malloc", "tf_free"))
    _TF = ctypes.CDLL(_so)
    _TF.tf_sweep.restype = ctypes.c_long
    _TF.tf_alloc_count.restype = ctypes.c_long
elif os.environ.get("SVSR_REDZONE") == "1":
    import ctypes
    import subprocess
    import tempfile

    import torch

    _src = os.path.join(ROOT, "tests", "redzone", "redzone_alloc.cpp")
    _so = os.path.join(tempfile.mkdtemp(prefix="svsr_rz_"), "libredzone.so")
    subprocess.run(["hipcc", "-shared", "-fPIC", "-O2", "-o", _so, _src], check=True, capture_output=True)
    torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(_so, "rz_malloc", "rz_free"))
    _RZ = ctypes.CDLL(_so)
    _RZ.rz_check_all.restype = ctypes.c_long
    _RZ.rz_alloc_count.restype = ctypes.c_long


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


@pytest.fixture(autouse=True)
def _redzone_sweep():
    """Under SVSR_REDZONE=1: fail the test that wrote outside a tensor it handed to the C ABI."""
    if _TF is not None:          # tail-flush mode: an out-of-bounds read does not reach this line (the process is gone); release what the test freed
        yield
        assert _TF.tf_sweep() == 0, "tf_malloc failed: the virtual-memory API is not available, this run proves nothing"
        assert _TF.tf_alloc_count() > 0
        return
    if _RZ is None:
        yield
        return
    before = _RZ.rz_check_all()
    yield
    after = _RZ.rz_check_all()
    assert after == before, f"{after - before} byte(s) were written into the red zones around device tensors during this test (stderr names the tensor sizes)"
