import os, sys, ctypes, subprocess, tempfile, torch
ROOT = os.getcwd()
_so = os.path.join(tempfile.mkdtemp(prefix="svsr_tf_"), "libredzone.so")
subprocess.run(["hipcc", "-shared", "-fPIC", "-O2", "-w", "-o", _so, os.path.join(ROOT, "tests/redzone/redzone_alloc.cpp")], check=True)
torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(_so, "tf_malloc", "tf_free"))
for n in (5, 100, 1000, 5000, 100000, 3000000):
    a = torch.arange(n, dtype=torch.int32)
    b = a.to("cuda")
    c = torch.empty(n, dtype=torch.int32).pin_memory(); c.copy_(a)
    d = c.to("cuda", non_blocking=True)
    e = torch.arange(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    print(n, "pageable H2D ok:", bool((b.cpu() == a).all()), " pinned H2D ok:", bool((d.cpu() == a).all()), " device arange ok:", bool((e.cpu() == a).all()),
          " d2d:", bool((b + 0 == e).all().item()), hex(b.data_ptr()))
sys.path.insert(0, ROOT)
from syncvsr_amd import ops
import math
TF = ctypes.CDLL(_so); TF.tf_sweep.restype = ctypes.c_long
def conv(N, H, W, Ci, Co, k, s_, p_):
    x = (torch.randn(N, H, W, Ci) * 0.5).to(torch.bfloat16)
    w = (torch.randn(Co, k, k, Ci) / math.sqrt(k * k * Ci)).to(torch.bfloat16)
    out, st = ops.conv2d_fwd(x.cuda(), w.cuda(), k, s_, p_, want_stats=True)
    torch.cuda.synchronize()
    return float(out.float().abs().max())
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
if mode in ("both", "nosweep"):
    print("c64 case:", conv(3, 11, 11, 64, 64, 3, 1, 1))
    if mode == "both":
        print("sweep ->", TF.tf_sweep())
print("generic case:", conv(2, 12, 12, 64, 128, 3, 2, 1))
print("generic again:", conv(2, 12, 12, 64, 128, 3, 2, 1))
print("sweep ->", TF.tf_sweep())
print("generic after sweep:", conv(2, 12, 12, 64, 128, 3, 2, 1))
print("other generic:", conv(2, 11, 11, 64, 128, 3, 2, 1))
