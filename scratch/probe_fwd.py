"""Times the segments of k_igemm_fwd_glds's K loop with s_memtime (build with -DSVSR_PROBE)."""
import ctypes, math, os, sys, torch
sys.path.insert(0, '.')
from syncvsr_amd import ops, _lib
dev = torch.device('cuda:0')
BF = torch.bfloat16
N = 928
for name, hw, ci, co in (("L2", 11, 128, 128), ("L3", 6, 256, 256), ("L4", 3, 512, 512)):
    x = torch.randn(N, hw, hw, ci, device=dev).to(BF)
    w = (torch.randn(co, 3, 3, ci, device=dev) / math.sqrt(9 * ci)).to(BF)
    for _ in range(3):
        out, st = ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=True)
    torch.cuda.synchronize()
    buf = torch.zeros(8, dtype=torch.int64, device=dev)
    lib = _lib.load()
    lib.svsr_probe_set.argtypes = [ctypes.c_void_p]
    lib.svsr_probe_set(buf.data_ptr())
    out, st = ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=True)
    torch.cuda.synchronize()
    b = buf.cpu().tolist()
    it = max(b[4], 1)
    print(name, "iters", b[4], "cycles/iter: wait+barrier %.0f  stage-issue %.0f  mma %.0f  | prologue %d epilogue %d total %d" % (b[0] / it, b[1] / it, b[2] / it, b[3], b[5], b[6]))
    lib.svsr_probe_set(None)
