#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_shapes.py -k "stem" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/stemfwd_tests.log 2>&1
echo "exit $?" >> gpurun_out/stemfwd_tests.log
grep -E "passed|failed|^FAILED|^ERROR|Error|assert|exit" gpurun_out/stemfwd_tests.log | tail -10
python - <<'PY' 2>&1 | tail -4
import torch
import syncvsr_amd
from syncvsr_amd import ops
dev = torch.device("cuda:0")
vid = torch.randn(32, 1, 29, 88, 88, device=dev)
w = torch.randn(64, 1, 5, 7, 7, device=dev) * 0.05
def run(dma, n=20):
    ops.tune("stem_fwd_dma", dma)
    for _ in range(3): ops.stem_conv_fwd(vid, w, want_stats=True)
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): ops.stem_conv_fwd(vid, w, want_stats=True)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
a, st_a = ops.stem_conv_fwd(vid, w, want_stats=True); ops.tune("stem_fwd_dma", 0); b, st_b = ops.stem_conv_fwd(vid, w, want_stats=True)
print("bit-identical outputs:", torch.equal(a, b))
for dma in (0, 1, 0, 1): print("stem_fwd_dma", dma, f"{run(dma):.1f} us (prep + conv)")
PY
bash scripts/gpu_bench_ab.sh "stem_fwd_dma=0" "stem_fwd_dma=1"
