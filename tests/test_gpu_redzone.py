"""Memory-safety pass over the kernels (-m gpu; SURVEY.md section 5 "race detection / sanitizers"): the kernel-level test files are run
again in a child pytest process whose device allocator surrounds EVERY tensor with poisoned 4 KiB red zones (tests/redzone/
redzone_alloc.cpp through tests/conftest.py, SVSR_REDZONE=1); a sweep after each test fails the test whose launches wrote into one.  What
this covers: the hand-computed LDS-DMA source offsets, swizzles, per-lane 32-bit offsets, padding rows of partial tiles and workspace
sizes of every C-ABI entry point those files exercise — including the benchmark-batch shapes and the fused encoder's cluster launches."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FILES = ["test_gpu_kernels.py", "test_gpu_bench_shapes.py", "test_gpu_enc_fused.py", "test_gpu_lrs_kernels.py"]


def test_kernel_tests_pass_with_red_zones_around_every_tensor():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ, SVSR_REDZONE="1", PYTHONPATH=os.pathsep.join([ROOT, HERE, os.environ.get("PYTHONPATH", "")]))
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", *[os.path.join(HERE, f) for f in FILES]]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=3000)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "redzone" not in r.stderr, tail
    print(r.stdout.strip().splitlines()[-1])
