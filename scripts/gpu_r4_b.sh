#!/bin/bash
# unit-list weight gradients: parity tests, per-conv timings old vs new, step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_shapes.py tests/test_gpu_lrs_kernels.py -x -q -k "wgrad or weight or linear or bench" 2>&1 | tail -5
for t in wg_units=0 wg_units=1; do
python - $t <<'PY'
import sys, os
sys.argv = [sys.argv[0]] + sys.argv[1:]
tag = sys.argv[1]
sys.path.insert(0, os.getcwd())
import torch
from syncvsr_amd import ops
k, v = tag.split("="); ops.tune(k, int(v))
exec(open("scripts/probes/trunk_times.py").read().split("for name, (H, W, Ci, Co, k, s, p) in TRUNK.items():")[0])
print(tag)
for name, (H, W, Ci, Co, k, s, p) in TRUNK.items():
    Ho, Wo = ops.conv_out_size(H, k, s, p), ops.conv_out_size(W, k, s, p)
    x = (torch.randn(N, H, W, Ci, device=dev) * 0.5).to(BF16)
    dy = (torch.randn(N, Ho, Wo, Co, device=dev) * 0.5).to(BF16)
    dw = torch.zeros(Co, k, k, Ci, device=dev)
    flops = 2.0 * N * Ho * Wo * Co * Ci * k * k
    t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, k, s, p))
    ops.HALO_WGRAD = False
    t_g = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, k, s, p))
    ops.HALO_WGRAD = True
    pl = ops.wgrad_conv_plan(N, H, W, Ci, Co, k, s, p)
    print(f"  {name:22s} {flops / 1e9:6.1f} GF | wgrad {t_w:6.1f} us {flops / t_w / 1e6:5.0f} TF | generic {t_g:6.1f} us | {pl.label} meta {list(pl.meta)}")
PY
done
for t in wg_units=0 wg_units=1 wg_units=0 wg_units=1; do
  timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-lrs-leg --profile-steps 1 --tune $t > gpurun_out/r4b/b_$t.json 2> gpurun_out/r4b/b_$t.err
  python - $t <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r4b/b_{t}.json").read().strip().splitlines()[-1]); print(t, d["ms_per_step"], "host", d.get("host_enqueue_ms"), {k: v["ms_per_step"] for k, v in d["roofline"]["per_kernel"].items() if "wgrad" in k})
except Exception as e: print(t, "FAILED", e)
PY
done
