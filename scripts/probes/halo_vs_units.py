import sys, math, torch
sys.path.insert(0, '.')
from syncvsr_amd import ops
dev = torch.device('cuda:0'); BF = torch.bfloat16
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for N in (928, 2560):
    for (H, C) in ((22, 64), (11, 128)):
        x = (torch.randn(N, H, H, C, device=dev) * 0.5).to(BF); dy = (torch.randn(N, H, H, C, device=dev) * 0.5).to(BF)
        dw = torch.zeros(C, 3, 3, C, device=dev)
        r = []
        for halo in (True, False):
            ops.HALO_WGRAD = halo
            r.append(t(lambda: ops.conv2d_wgrad(x, dy, dw, 3, 1, 1)))
        print(f"N={N} {H}x{H} C={C}: halo {r[0]:.1f} us, per-tap units {r[1]:.1f} us")
