import ctypes, math, sys, torch
sys.path.insert(0, '.')
from syncvsr_amd import ops, _lib
dev = torch.device('cuda:0'); BF = torch.bfloat16
lib = _lib.load(); lib.svsr_probe_set.argtypes = [ctypes.c_void_p]
buf = torch.zeros(8, dtype=torch.int64, device=dev)
def run(name, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); buf.zero_(); lib.svsr_probe_set(buf.data_ptr()); fn(); torch.cuda.synchronize(); lib.svsr_probe_set(None)
    b = buf.cpu().tolist()
    print(f"{name:28s} stats {b[0]:6d}  acc->lds {b[1]:6d}  sync {b[2]:6d}  store-loop {b[3]:6d}  | epilogue total {b[4]:6d}")
N = 928
for name, hw, ci, co in (("L2", 11, 128, 128), ("L3", 6, 256, 256)):
    x = torch.randn(N, hw, hw, ci, device=dev).to(BF); w = (torch.randn(co, 3, 3, ci, device=dev) / math.sqrt(9 * ci)).to(BF)
    run(name + " fwd+stats", lambda: ops.conv2d_fwd(x, w, 3, 1, 1, want_stats=True))
    run(name + " fwd", lambda: ops.conv2d_fwd(x, w, 3, 1, 1))
    dy = torch.randn(N, hw, hw, co, device=dev).to(BF); wt = w.permute(3, 1, 2, 0).contiguous(); add = torch.randn(N, hw, hw, ci, device=dev).to(BF)
    run(name + " dgrad+addend", lambda: ops.conv2d_dgrad(dy, wt, 3, 1, 1, (hw, hw), addend=add))
R = 960
for nm, K, Nn, gelu in (("qkv", 512, 1536, False), ("ffn1 gelu", 512, 2048, True), ("ffn2", 2048, 512, False)):
    x = torch.randn(R, K, device=dev).to(BF); w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).to(BF); b = torch.zeros(Nn, device=dev)
    run("linear " + nm, lambda: ops.linear_fwd(x, w, b, rows=R, K=K, N=Nn, x_pitch=K, gelu=gelu))
