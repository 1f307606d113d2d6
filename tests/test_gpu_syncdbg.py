"""Latent-race check of the LDS-DMA pipelines (-m gpu; SURVEY.md section 5 "race detection").  The contraction kernels keep tiles in flight across
barriers with COUNTED waits (s_waitcnt vmcnt(n), n > 0, in igemm_p8.hip, igemm_fwd.hip, igemm_wgrad.hip, enc_fused.hip, audio_head.hip).  A
count that is one piece short, or a ring slot re-staged a phase early, reads LDS bytes the DMA has not written yet — and still passes every
parity test whenever the DMA happens to land first.  The build variant `syncdbg` (-DSVSR_SYNC_DEBUG, csrc/common.h; made by
__graft_entry__.build()) turns every counted wait into vmcnt(0) and runs the persistent kernel's wave groups in lock step: nothing is in
flight when anything is read.  Both libraries run the same forward + backward of both models at the benchmark shapes in child processes;
every output, statistic and parameter gradient must agree BIT FOR BIT — the arithmetic is the same, only the waiting differs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run(variant: str, out: str) -> dict:
    env = dict(os.environ, SVSR_LIB_VARIANT=variant, PYTHONPATH=os.pathsep.join([ROOT, os.environ.get("PYTHONPATH", "")]))
    if not variant:
        env.pop("SVSR_LIB_VARIANT")
    r = subprocess.run([sys.executable, os.path.join(HERE, "syncdbg_worker.py"), out], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-3000:]
    return json.load(open(out))


def test_counted_waits_give_the_same_bits_as_full_waits(tmp_path):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from syncvsr_amd import build

    dbg = build.lib_path("syncdbg")
    if not os.path.exists(dbg) or build.stale("syncdbg"):
        build.build(verbose=False, variant="syncdbg")
    a = _run("", str(tmp_path / "product.json"))
    b = _run("syncdbg", str(tmp_path / "syncdbg.json"))
    assert a.pop("library") == "libsyncvsr_hip.so" and b.pop("library") == "libsyncvsr_hip_syncdbg.so"
    assert set(a) == set(b) and len(a) > 400, (len(a), len(b))
    diff = sorted(k for k in a if a[k] != b[k])
    assert not diff, f"{len(diff)} of {len(a)} tensors differ between counted waits and full waits (a latent race?): {diff[:12]}"
    print(f"{len(a)} tensors bit-identical between the product library and the full-wait build")
