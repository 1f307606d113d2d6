import ctypes, math, sys, torch
sys.path.insert(0, '.')
from syncvsr_amd import ops, _lib
dev = torch.device('cuda:0'); BF = torch.bfloat16
x = torch.randn(928, 22, 22, 64, device=dev).to(BF)
w = (torch.randn(64, 3, 3, 64, device=dev) / 24).to(BF)
for _ in range(3): ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=True)
torch.cuda.synchronize()
buf = torch.zeros(8, dtype=torch.int64, device=dev)
lib = _lib.load(); lib.svsr_probe64_set.argtypes = [ctypes.c_void_p]
lib.svsr_probe64_set(buf.data_ptr())
ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=True); torch.cuda.synchronize()
b = buf.cpu().tolist(); n = max(b[5], 1)
print("c64 chunks", b[5], "cycles/chunk: top-barrier %.0f stage %.0f mfma %.0f mid-barrier %.0f epilogue %.0f | total %d" % (b[0]/n, b[1]/n, b[2]/n, b[3]/n, b[4]/n, b[6]))
