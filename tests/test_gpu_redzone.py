"""Memory-safety pass over the kernels (-m gpu; SURVEY.md section 5 "race detection / sanitizers"): the kernel-level test files are run
again in a child pytest process whose device allocator surrounds EVERY tensor with poisoned 4 KiB red zones (tests/redzone/
redzone_alloc.cpp through tests/conftest.py, SVSR_REDZONE=1); a sweep after each test fails the test whose launches wrote into one.  What
this covers: the hand-computed LDS-DMA source offsets, swizzles, per-lane 32-bit offsets, padding rows of partial tiles and workspace
sizes of every C-ABI entry point those files exercise — including the benchmark-batch shapes and the fused encoder's cluster launches.

Red zones see STORES only.  The second pass (SVSR_TAILFLUSH=1, same allocator source) covers out-of-bounds READS: every tensor ends flush (to the
kernels' 16-byte vector width) against an unmapped page, so a per-lane offset that reads behind a tensor, or a padding row fetched from behind
the last image, is a GPU memory-access fault that kills the child process in the test that did it.  tailflush_selftest proves the mechanism
first: the last 16 bytes of a tensor read fine, the next 16 abort the process."""
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


def _hipcc(out, *srcs):
    subprocess.run(["hipcc", "-O2", "-w", "--offload-arch=gfx950", "-o", out, *srcs], check=True, capture_output=True, timeout=600)


def test_tail_flush_allocator_faults_on_a_read_behind_a_tensor(tmp_path):
    """The mechanism of the pass below, proven on a 16-byte read: inside the tensor -> fine; 16 bytes behind it -> the process dies."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    exe = str(tmp_path / "tailflush_selftest")
    _hipcc(exe, os.path.join(HERE, "redzone", "tailflush_selftest.cpp"), os.path.join(HERE, "redzone", "redzone_alloc.cpp"))
    ok = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0 and "inside ok" in ok.stdout, ok.stdout + ok.stderr
    over = subprocess.run([exe, "over"], capture_output=True, text=True, timeout=120)
    assert "inside ok" in over.stdout and "NOT caught" not in over.stdout and over.returncode != 0, (over.returncode, over.stdout, over.stderr[-800:])
    print(f"read behind the tensor: exit {over.returncode}: {(over.stderr.strip().splitlines() or ['(no message)'])[0][:160]}")


def test_kernel_tests_pass_with_every_tensor_flush_against_an_unmapped_page():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ, SVSR_TAILFLUSH="1", PYTHONPATH=os.pathsep.join([ROOT, HERE, os.environ.get("PYTHONPATH", "")]))
    env.pop("SVSR_REDZONE", None)
    cmd = [sys.executable, "-m", "pytest", "-v", "-x", "-m", "gpu", "-p", "no:cacheprovider", *[os.path.join(HERE, f) for f in FILES]]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=3000)
    lines = [ln for ln in r.stdout.splitlines() if "::" in ln]
    tail = "last test line: " + (lines[-1] if lines else "(none)") + "\n" + r.stdout[-1500:] + "\n" + r.stderr[-3000:]
    assert r.returncode == 0, tail            # (a memory-access fault aborts the child: the test named last read behind one of its tensors)
    assert " passed" in r.stdout, tail
    print(r.stdout.strip().splitlines()[-1])
