#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_shapes.py -x -q -k "wgrad or weight or bench" 2>&1 | tail -3
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from syncvsr_amd import ops
exec(open("scripts/probes/trunk_times.py").read().split("for name, (H, W, Ci, Co, k, s, p) in TRUNK.items():")[0])
for tag in ("w3_waves8=0", "w3_waves8=1", "w3_waves8=1,w3_blocks8=128", "w3_waves8=1,w3_blocks8=64"):
    for kv in tag.split(","):
        k_, v_ = kv.split("="); ops.tune(k_, int(v_))
    out = []
    for name in ("layer1.conv", "layer2.conv"):
        H, W, Ci, Co, k, s, p = TRUNK[name]
        x = (torch.randn(N, H, W, Ci, device=dev) * 0.5).to(BF16)
        dy = (torch.randn(N, H, W, Co, device=dev) * 0.5).to(BF16)
        dw = torch.zeros(Co, k, k, Ci, device=dev)
        flops = 2.0 * N * H * W * Co * Ci * 9
        t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, k, s, p))
        out.append(f"{name} {t_w:6.1f} us {flops / t_w / 1e6:5.0f} TF")
    print(tag, "|", " | ".join(out))
PY
for t in w3_waves8=0 w3_waves8=1 w3_waves8=0 w3_waves8=1; do
  timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-lrs-leg --profile-steps 1 --tune $t > gpurun_out/r4f/b.json 2> gpurun_out/r4f/b.err
  python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r4f/b.json").read().strip().splitlines()[-1]); print(t, d["ms_per_step"], "host", d.get("host_enqueue_ms"), "loss", d["final_loss"], {k: v["ms_per_step"] for k, v in d["roofline"]["per_kernel"].items() if "halo" in k})
except Exception as e: print(t, "FAILED", e); print(open("gpurun_out/r4f/b.err").read()[-1500:])
PY
done
